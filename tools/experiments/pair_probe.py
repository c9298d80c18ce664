#!/usr/bin/env python3
"""What would pairing two samples' 16 x 16 regions in one 16 x 32 workgroup buy on BASELINE config 2's deep layers?  The paired form has the cost of the
existing 16 x 32 split-K kernel on a (B / 2, 16, 32) image: timed here next to the 16 x 16-region split-K form on (B, 16, 16) -- same MFMAs, half the
weight-fragment stream per sample."""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
torch.zeros(1, device="cuda")
from noisediff_amd import _lib as L
import hiputil as hu
ctx = hu.Ctx()
REPS = 20


def timed(fn):
    e0, e1 = C.c_void_p(), C.c_void_p(); L.call("nd_event_create", C.byref(e0)); L.call("nd_event_create", C.byref(e1))
    fn(); ctx.sync()
    L.call("nd_event_record", e0, ctx.stream)
    for _ in range(REPS): fn()
    L.call("nd_event_record", e1, ctx.stream); ms = C.c_float(); L.call("nd_event_elapsed_ms", e0, e1, C.byref(ms))
    return ms.value / REPS * 1e3


def run(entry, B, H, W, cin, cout, splits, stats):
    x = torch.randn(B, H, W, cin, device=hu.DEV); w = torch.randn(cout, cin, 3, 3) * 0.05
    wd = hu.dev(w); wp = torch.empty(ctx.lib.nd_pack_conv3x3_wino4_weight_floats(cin, cout), device=hu.DEV)
    L.call("nd_pack_conv3x3_wino4_weight", wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream); ctx.sync()
    b = torch.randn(cout, device=hu.DEV); out = torch.empty(B, H, W, cout, device=hu.DEV)
    slots = ctx.lib.nd_conv3x3_wino4_stat_slots(H, W)
    st = torch.empty(B, slots, cout, 2, device=hu.DEV); sc = torch.empty(slots, device=hu.DEV)
    ws = torch.empty(ctx.lib.nd_conv3x3_wino4_splitk_workspace_floats(B, H, W, cout, splits), device=hu.DEV)
    torch.cuda.synchronize()
    d = L.Conv3x3(); d.src, d.weight, d.bias, d.out = hu.src(x), wp.data_ptr(), b.data_ptr(), out.data_ptr()
    if stats: d.stats, d.slot_count = st.data_ptr(), sc.data_ptr()
    d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
    return timed(lambda: L.call(entry, C.byref(d), ws.data_ptr(), splits, ctx.stream))


B = 16
for (S, cin, cout) in [(16, 256, 256), (16, 512, 512), (16, 768, 512), (16, 256, 512)]:
    s16 = ctx.lib.nd_conv3x3_wino4_16_splitk_plan(S, S, cin, cout)
    t16 = run("nd_conv3x3_wino4_16_splitk_nhwc_f32", B, S, S, cin, cout, s16, True)
    line = f"{cin}->{cout} @{S}x{S} x{B}: 16-form splits {s16}: {t16:6.1f} us |"
    for sp in (2, 4, 8):
        if (cin // 16) % sp == 0 and cin // 16 // sp >= 2:
            line += f" paired-cost (B/2, {S}, {2 * S}) 16x32-form splits {sp}: {run('nd_conv3x3_wino4_splitk_nhwc_f32', B // 2, S, 2 * S, cin, cout, sp, True):6.1f} us |"
    print(line, flush=True)
