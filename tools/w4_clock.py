"""Diagnostic: effective shader clock and cycle split of conv3x3_wino4 (needs a -DW4_STAMP side build, ND_LIB)."""
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
    wd = hu.dev(w); wp = torch.empty(ctx.lib.nd_pack_conv3x3_wino4_weight_floats(cin, cout), device=hu.DEV)
    L.call("nd_pack_conv3x3_wino4_weight", wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream); ctx.sync()
    out = torch.empty(B, H, W, cout, device=hu.DEV)
    dbg = torch.zeros(8 * 256, dtype=torch.int64, device=hu.DEV)
    st = torch.zeros(B * ctx.lib.nd_conv3x3_wino_stat_slots(H, W) * cout * 2, device=hu.DEV)
    torch.cuda.synchronize()
    d = L.Conv3x3(); d.src, d.weight, d.out = hu.src(x), wp.data_ptr(), out.data_ptr()
    d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
    d.slot_count = dbg.data_ptr(); d.stats = st.data_ptr()
    for _ in range(3):
        L.call("nd_conv3x3_wino4_nhwc_f32", C.byref(d), ctx.stream); ctx.sync()
    v = dbg.cpu().view(256, 8).double()
    cyc, real, chunks, epi, xf, wait, drain = (v[:, i] for i in range(7))
    n_chunks = (cin + 15) // 16
    mhz = cyc / (real / 100.0)
    per = lambda t: float((t / chunks).mean())
    print((B, H, W, cin, cout), f"clock {mhz.mean():.0f} MHz (min {mhz.min():.0f} max {mhz.max():.0f}); cycles/chunk {per(cyc):.0f} (MFMA 9216):",
          f"stages {per(cyc - epi - xf - wait):.0f}, wait+barrier {per(wait):.0f}, transform+barrier {per(xf):.0f}, epilogue {per(epi):.0f}"
          f" (= {float((epi / (chunks / n_chunks)).mean()):.0f} per tile), drain {float((drain / (chunks / n_chunks)).mean()):.0f} per tile; wall/WG {float(real.mean()) / 100:.1f} us", flush=True)
