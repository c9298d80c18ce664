"""Diagnostic: effective shader clock and cycles per K chunk of conv3x3_wino2 (needs a -DW2_STAMP side build, ND_LIB)."""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
torch.zeros(1, device="cuda")
from noisediff_amd import _lib as L
L.load(os.environ["ND_LIB"])
import hiputil as hu
ctx = hu.Ctx()
for (B, H, W, cin, cout) in [(16, 256, 256, 64, 64), (16, 32, 32, 512, 512)]:
    x = torch.randn(B, H, W, cin, device=hu.DEV); w = torch.randn(cout, cin, 3, 3) * 0.05
    wd = hu.dev(w); wp = torch.empty(ctx.lib.nd_pack_conv3x3_wino_weight_floats(cin, cout), device=hu.DEV)
    L.call("nd_pack_conv3x3_wino_weight", wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream); ctx.sync()
    out = torch.empty(B, H, W, cout, device=hu.DEV)
    dbg = torch.zeros(4 * 256, dtype=torch.int64, device=hu.DEV)
    torch.cuda.synchronize()
    d = L.Conv3x3(); d.src, d.weight, d.out = hu.src(x), wp.data_ptr(), out.data_ptr()
    d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
    d.slot_count = dbg.data_ptr(); d.stats = 0
    for _ in range(3):
        L.call("nd_conv3x3_wino2_nhwc_f32", C.byref(d), ctx.stream); ctx.sync()
    v = dbg.cpu().view(256, 4).double()
    cyc, real, chunks, epi = v[:, 0], v[:, 1], v[:, 2], v[:, 3]
    n_chunks = (cin + 31) // 32
    mhz = cyc / (real / 100.0)
    print((B, H, W, cin, cout), f"clock {mhz.mean():.0f} MHz (min {mhz.min():.0f} max {mhz.max():.0f});",
          f"cycles/chunk {float((cyc / chunks).mean()):.0f} (ideal 16384); epilogue cycles/tile {float((epi / (chunks / n_chunks)).mean()):.0f};",
          f"K-loop cycles/chunk {float(((cyc - epi) / chunks).mean()):.0f}; wall/WG {float(real.mean()) / 100:.1f} us", flush=True)
