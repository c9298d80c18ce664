"""Diagnostic: cycle split of conv3x3_f16x3 (needs a -DD16_STAMP side build: tools/variant.sh conv3x3_f16x3 stampd -DD16_STAMP; ND_LIB=tools/_build/lib_stampd.so)."""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
torch.zeros(1, device="cuda")
from noisediff_amd import _lib as L
L.load(os.environ["ND_LIB"])
import hiputil as hu
ctx = hu.Ctx()
SHAPES = [(16, 256, 256, 64, 64, 0), (16, 256, 256, 64, 64, 1), (16, 128, 128, 128, 128, 0), (16, 32, 32, 512, 512, 0)]
for (B, H, W, cin, cout, mode) in SHAPES:
    x = torch.randn(B, H, W, cin, device=hu.DEV); w = torch.randn(cout, cin, 3, 3) * 0.05
    wd = hu.dev(w); wp = torch.empty(ctx.lib.nd_pack_conv3x3_f16x3_weight_floats(cin, cout), device=hu.DEV)
    L.call("nd_pack_conv3x3_f16x3_weight", wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream); ctx.sync()
    out = torch.empty(B, H, W, cout, device=hu.DEV)
    dbg = torch.zeros(8 * 1024, dtype=torch.int64, device=hu.DEV)
    st = torch.zeros(B * ctx.lib.nd_conv3x3_wino4_stat_slots(H, W) * cout * 2, device=hu.DEV)
    mad = torch.rand(B, 3, cin, device=hu.DEV) + 0.5
    torch.cuda.synchronize()
    d = L.Conv3x3(); d.src, d.weight, d.out = hu.src(x, None, mode, **({"mad": mad} if mode else {})), wp.data_ptr(), out.data_ptr()
    d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
    d.slot_count = dbg.data_ptr(); d.stats = st.data_ptr()
    for _ in range(3):
        L.call("nd_conv3x3_f16x3_nhwc_f32", C.byref(d), ctx.stream); ctx.sync()
    v = dbg.cpu().view(1024, 8).double()
    v = v[v[:, 5] > 0]
    cyc, real, mma, bar, epi, tiles = (v[:, i] for i in range(6))
    n_chunks = cin // 16
    span = (float(v[:, 7].max()) - float(v[:, 6].min())) / 100.0          # first workgroup's start to the last one's end (100 MHz counter)
    late = (v[:, 6] - v[:, 6].min()) / 100.0
    mhz = cyc / (real / 100.0)
    pt = lambda t: float((t / tiles).mean())
    print((B, H, W, cin, cout, mode), f"{v.shape[0]} workgroups, clock {mhz.mean():.0f} MHz, wall/WG {float(real.mean()) / 100:.1f} us (min {float(real.min()) / 100:.1f}, max {float(real.max()) / 100:.1f}), {float(tiles.mean()):.1f} tiles x {n_chunks} chunks;",
          f"span {span:.1f} us, starts spread over {float(late.max()):.1f} us (median {float(late.median()):.1f}); per tile: {pt(cyc):.0f} cycles (MFMA issue {216 * 33.7 * n_chunks:.0f}) = chunks {pt(mma):.0f} ({pt(mma) / n_chunks:.0f} each), barrier waits {pt(bar):.0f}, epilogue {pt(epi):.0f}, rest {pt(cyc - mma - bar - epi):.0f}", flush=True)
