"""Stress check for the inline-asm MFMA kernel: the same conv launched 12 times per shape must be bitwise repeatable and finite
(a missed VALU->MFMA or MFMA->read hazard would show up as run-to-run differences).  Run on the MI355X box."""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
torch.zeros(1, device="cuda")
from noisediff_amd import _lib as L
import hiputil as hu
ctx = hu.Ctx()
bad = 0
for (B, H, W, cin, cout, mode) in [(16, 256, 256, 64, 64, 0), (16, 256, 256, 64, 64, 1), (16, 32, 32, 512, 512, 0), (4, 128, 128, 128, 64, 1), (2, 36, 52, 96, 40, 0)]:
    x = torch.randn(B, H, W, cin, device=hu.DEV); w = torch.randn(cout, cin, 3, 3) * 0.05
    wd = hu.dev(w); wp = torch.empty(ctx.lib.nd_pack_conv3x3_wino_weight_floats(cin, cout), device=hu.DEV)
    L.call("nd_pack_conv3x3_wino_weight", wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream); ctx.sync()
    b = torch.randn(cout, device=hu.DEV); mad = torch.rand(B, 3, cin, device=hu.DEV) + 0.5
    slots = ctx.lib.nd_conv3x3_wino_stat_slots(H, W)
    outs = []
    for rep in range(12):
        out = torch.full((B, H, W, cout), float("nan"), device=hu.DEV)
        st = torch.full((B, slots, cout, 2), float("nan"), device=hu.DEV); sc = torch.empty(slots, device=hu.DEV)
        torch.cuda.synchronize()
        d = L.Conv3x3(); d.src, d.weight, d.bias, d.out = hu.src(x, None, mode, mad=mad), wp.data_ptr(), b.data_ptr(), out.data_ptr()
        d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
        d.stats, d.slot_count = st.data_ptr(), sc.data_ptr()
        L.call("nd_conv3x3_wino2_nhwc_f32", C.byref(d), ctx.stream); ctx.sync()
        outs.append((out.clone(), st.clone()))
    same = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
    fin = bool(torch.isfinite(outs[0][0]).all() and torch.isfinite(outs[0][1]).all())
    print((B, H, W, cin, cout, mode), "bitwise repeatable:", same, "finite:", fin, flush=True)
    bad += (not same) or (not fin)
print("STRESS", "FAILED" if bad else "OK")
