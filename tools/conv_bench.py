import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
import hiputil as hu
from noisediff_amd import _lib as L
ctx = hu.Ctx()
def bench(B, H, W, cin, cout, mode=0, stats=True, reps=5):
    x = torch.randn(B, H, W, cin, device=hu.DEV); w = torch.randn(cout, cin, 3, 3) * 0.05
    wp = hu.pack_conv3(ctx, w); b = torch.randn(cout, device=hu.DEV)
    mad = torch.rand(B, 3, cin, device=hu.DEV) + 0.5
    out = torch.empty(B, H, W, cout, device=hu.DEV)
    slots = ctx.lib.nd_conv3x3_stat_slots(H, W, cout, B)
    st = torch.empty(B, slots, cout, 2, device=hu.DEV); sc = torch.empty(slots, device=hu.DEV)
    torch.cuda.synchronize()
    s = hu.src(x, None, mode, mad=mad)
    d = L.Conv3x3(); d.src, d.weight, d.bias, d.out = s, wp.data_ptr(), b.data_ptr(), out.data_ptr()
    d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
    if stats: d.stats, d.slot_count = st.data_ptr(), sc.data_ptr()
    e0, e1 = C.c_void_p(), C.c_void_p(); L.call("nd_event_create", C.byref(e0)); L.call("nd_event_create", C.byref(e1))
    L.call("nd_conv3x3_nhwc_f32", C.byref(d), ctx.stream); ctx.sync()
    L.call("nd_event_record", e0, ctx.stream)
    for _ in range(reps): L.call("nd_conv3x3_nhwc_f32", C.byref(d), ctx.stream)
    L.call("nd_event_record", e1, ctx.stream); ms = C.c_float(); L.call("nd_event_elapsed_ms", e0, e1, C.byref(ms))
    t = ms.value / reps
    return t * 1e3, 18.0 * cin * cout * H * W * B / t / 1e9, ctx.lib.nd_conv3x3_tiling_id(B, H, W, cout)
shapes = [(16, 256, 256, 64, 64), (16, 256, 256, 128, 64), (16, 128, 128, 128, 128), (16, 64, 64, 256, 256), (16, 32, 32, 512, 512), (16, 32, 32, 768, 512)]
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
for sh in shapes:
    for mode in (0, 1):
        us, tf, til = bench(*sh, mode=mode)
        print(f"{sh} mode={mode} t{til}: {us:8.1f} us {tf:6.1f} TF", flush=True)
