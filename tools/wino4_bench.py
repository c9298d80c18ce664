#!/usr/bin/env python3
"""conv3x3_wino4 (experimental F(4x4,3x3)) against conv3x3_wino2 on the benchmark's representative shapes: us and algorithmic TF."""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
torch.zeros(1, device="cuda")
from noisediff_amd import _lib as L
if os.environ.get("ND_LIB"):
    L.load(os.environ["ND_LIB"])
import hiputil as hu
ctx = hu.Ctx()

def bench(entry, pack, B, H, W, cin, cout, reps=5, mode=0, stats=False):
    x = torch.randn(B, H, W, cin, device=hu.DEV); w = torch.randn(cout, cin, 3, 3) * 0.05
    wd = hu.dev(w); wp = torch.empty(getattr(ctx.lib, pack + "_floats")(cin, cout), device=hu.DEV)
    L.call(pack, wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream); ctx.sync()
    b = torch.randn(cout, device=hu.DEV)
    out = torch.empty(B, H, W, cout, device=hu.DEV)
    torch.cuda.synchronize()
    mad = torch.rand(B, 3, cin, device=hu.DEV) + 0.5
    slots = ctx.lib.nd_conv3x3_wino_stat_slots(H, W)
    st = torch.empty(B, slots, cout, 2, device=hu.DEV); sc = torch.empty(slots, device=hu.DEV)
    torch.cuda.synchronize()
    d = L.Conv3x3(); d.src, d.weight, d.bias, d.out = hu.src(x, None, mode, mad=mad), wp.data_ptr(), b.data_ptr(), out.data_ptr()
    if stats: d.stats, d.slot_count = st.data_ptr(), sc.data_ptr()
    d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
    e0, e1 = C.c_void_p(), C.c_void_p(); L.call("nd_event_create", C.byref(e0)); L.call("nd_event_create", C.byref(e1))
    L.call(entry, C.byref(d), ctx.stream); ctx.sync()
    L.call("nd_event_record", e0, ctx.stream)
    for _ in range(reps): L.call(entry, C.byref(d), ctx.stream)
    L.call("nd_event_record", e1, ctx.stream); ms = C.c_float(); L.call("nd_event_elapsed_ms", e0, e1, C.byref(ms))
    t = ms.value / reps
    return t * 1e3, 18.0 * cin * cout * H * W * B / t / 1e9, out

for sh in [(16, 256, 256, 64, 64), (16, 256, 256, 128, 64), (16, 64, 64, 256, 256), (16, 32, 32, 512, 512)]:
    u2, t2, o2 = bench("nd_conv3x3_wino2_nhwc_f32", "nd_pack_conv3x3_wino_weight", *sh)
    u4, t4, o4 = bench("nd_conv3x3_wino4_nhwc_f32", "nd_pack_conv3x3_wino4_weight", *sh)
    print(sh, f"wino2 {u2:8.1f} us {t2:6.1f} TF | wino4 {u4:8.1f} us {t4:6.1f} TF | x{u2 / u4:.2f}", flush=True)
for sh in [(16, 256, 256, 64, 64), (16, 128, 128, 128, 128)]:
    u2, t2, o2 = bench("nd_conv3x3_wino2_nhwc_f32", "nd_pack_conv3x3_wino_weight", *sh, mode=1, stats=True)
    u4, t4, o4 = bench("nd_conv3x3_wino4_nhwc_f32", "nd_pack_conv3x3_wino4_weight", *sh, mode=1, stats=True)
    print(sh, f"affine+SiLU prologue, stats: wino2 {u2:8.1f} us {t2:6.1f} TF | wino4 {u4:8.1f} us {t4:6.1f} TF | x{u2 / u4:.2f}", flush=True)
