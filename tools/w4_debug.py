#!/usr/bin/env python3
"""Where is nd_conv3x3_wino4_nhwc_f32 wrong?  Error map per (sample, 16x16 tile, 16-cout group) on a few shapes."""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch, torch.nn.functional as F
from noisediff_amd import _lib as L
if os.environ.get("ND_LIB"):
    L.load(os.environ["ND_LIB"])
import hiputil as hu
ctx = hu.Ctx()
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(1, 32, 32, 16, 64), (1, 32, 32, 64, 64), (2, 64, 64, 64, 64)]
for (B, H, W, cin, cout) in shapes:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, cin, H, W, generator=g); w = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5; b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, w, b, padding=1)
    wd = hu.dev(w); wp = torch.empty(ctx.lib.nd_pack_conv3x3_wino4_weight_floats(cin, cout), device=hu.DEV)
    L.call("nd_pack_conv3x3_wino4_weight", wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream); ctx.sync()
    out = hu.full((B, H, W, cout)); bd = hu.dev(b)
    d = L.Conv3x3(); d.src, d.weight, d.bias, d.out = hu.src(hu.nhwc(x)), wp.data_ptr(), bd.data_ptr(), out.data_ptr()
    d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
    L.call("nd_conv3x3_wino4_nhwc_f32", C.byref(d), ctx.stream); ctx.sync()
    err = (hu.nchw(out) - ref).abs()
    print((B, H, W, cin, cout), "max err", float(err.max()), "nan", int(torch.isnan(hu.nchw(out)).sum()))
    for bb in range(B):
        for ty in range((H + 15) // 16):
            row = []
            for tx in range((W + 15) // 16):
                e = err[bb, :, ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16]
                row.append("[" + " ".join(f"{float(e[c:c + 16].max()):.0e}" for c in range(0, cout, 16)) + "]")
            print(f" b{bb} ty{ty}: " + "  ".join(row))
    # inside the first bad tile: per 4x4 tile
    bad = (err > 1e-3).nonzero()
    if len(bad):
        bb, c, y, xx = [int(v) for v in bad[0]]
        ty, tx = y // 16, xx // 16
        e = err[bb, :, ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16].amax(0)
        print(f" first bad: b{bb} c{c} y{y} x{xx}; 4x4-tile max of that 16x16 tile:")
        for i in range(4):
            print("   " + " ".join(f"{float(e[4 * i:4 * i + 4, 4 * j:4 * j + 4].max()):.0e}" for j in range(4)))

# ---- which patch row / column does the kernel get wrong?  CPU emulation of F(4x4,3x3) for one 4x4 tile with candidate corruptions
import numpy as np
BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)
B, H, W, cin, cout = 1, 32, 32, 16, 64
g = torch.Generator().manual_seed(1)
x = torch.randn(B, cin, H, W, generator=g); w = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5; b = torch.randn(cout, generator=g)
wd = hu.dev(w); wp = torch.empty(ctx.lib.nd_pack_conv3x3_wino4_weight_floats(cin, cout), device=hu.DEV)
L.call("nd_pack_conv3x3_wino4_weight", wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream); ctx.sync()
out = hu.full((B, H, W, cout)); bd = hu.dev(b)
d = L.Conv3x3(); d.src, d.weight, d.bias, d.out = hu.src(hu.nhwc(x)), wp.data_ptr(), bd.data_ptr(), out.data_ptr()
d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
L.call("nd_conv3x3_wino4_nhwc_f32", C.byref(d), ctx.stream); ctx.sync()
got = hu.nchw(out).double().numpy()
xp = F.pad(x, (1, 1, 1, 1)).double().numpy()[0]
U = np.einsum("ir,ocrs,js->ocij", G, w.double().numpy(), G)
def emulate(py, px, co, mod):
    dpatch = xp[:, py:py + 6, px:px + 6].copy()            # patch of the tile whose outputs start at (py, px): padded coords
    dpatch = mod(dpatch, py, px)
    V = np.einsum("ia,cab,jb->cij", BT, dpatch, BT)
    M = (U[co] * V).sum(0)
    return AT @ M @ AT.T + float(b[co])
cands = {"exact": lambda p, y, x_: p,
         "row5=0": lambda p, y, x_: np.concatenate([p[:, :5], np.zeros_like(p[:, :1])], 1),
         "row5=row4": lambda p, y, x_: np.concatenate([p[:, :5], p[:, 4:5]], 1),
         "row5=row2": lambda p, y, x_: np.concatenate([p[:, :5], p[:, 2:3]], 1),
         "rows345=rows012": lambda p, y, x_: np.concatenate([p[:, :3], p[:, :3]], 1),
         "rows345=0": lambda p, y, x_: np.concatenate([p[:, :3], np.zeros_like(p[:, :3])], 1)}
for (py, px) in [(0, 0), (4, 8), (16, 20)]:
    for co in (0, 37):
        tile = got[0, co, py:py + 4, px:px + 4]
        print(f"tile at ({py},{px}) cout {co}: " + "  ".join(f"{k}: {np.abs(emulate(py, px, co, f) - tile).max():.1e}" for k, f in cands.items()))
