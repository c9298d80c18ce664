#!/usr/bin/env python3
"""Correctness of every tools/_build/libw2nop_*.so side build (tools/w2_nop_probe.sh) on two shapes and two prologue modes:
relative error of nd_conv3x3_wino2_nhwc_f32 against F.conv2d.  Each library runs in its own process (ND_LIB)."""
import glob, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("ND_LIB"):
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    import ctypes as C
    import torch, torch.nn.functional as F
    from noisediff_amd import _lib as L
    L.load(os.environ["ND_LIB"])
    import hiputil as hu
    ctx = hu.Ctx()
    res = []
    for (B, H, W, cin, cout, mode) in [(2, 32, 32, 64, 64, 0), (1, 32, 32, 512, 512, 0), (2, 64, 64, 64, 64, 1), (1, 48, 48, 96, 128, 0)]:
        g = torch.Generator().manual_seed(1)
        x = torch.randn(B, cin, H, W, generator=g); w = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5; b = torch.randn(cout, generator=g)
        mad = torch.rand(B, 3, cin, generator=g) + 0.5
        xin = F.silu((x - mad[:, 0, :, None, None]) * mad[:, 1, :, None, None] + mad[:, 2, :, None, None]) if mode else x
        ref = F.conv2d(xin, w, b, padding=1)
        wd = hu.dev(w); wp = torch.empty(ctx.lib.nd_pack_conv3x3_wino_weight_floats(cin, cout), device=hu.DEV)
        L.call("nd_pack_conv3x3_wino_weight", wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream); ctx.sync()
        out = hu.full((B, H, W, cout)); bd = hu.dev(b)
        d = L.Conv3x3(); d.src, d.weight, d.bias, d.out = hu.src(hu.nhwc(x), None, mode, mad=hu.dev(mad)), wp.data_ptr(), bd.data_ptr(), out.data_ptr()
        d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
        L.call("nd_conv3x3_wino2_nhwc_f32", C.byref(d), ctx.stream); ctx.sync()
        err = float((hu.nchw(out) - ref).abs().max() / ref.abs().max())
        res.append(f"{err:.1e}")
    print(os.path.basename(os.environ["ND_LIB"]), " ".join(res), "OK" if all(float(r) < 1e-4 for r in res) else "WRONG", flush=True)
else:
    for lib in sorted(glob.glob(os.path.join(REPO, "tools", "_build", "libw2nop_*.so"))):
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, ND_LIB=lib), timeout=300)
