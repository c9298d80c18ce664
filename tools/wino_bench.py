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
ENTRY = os.environ.get("ND_WINO_ENTRY", "nd_conv3x3_wino_nhwc_f32")
def bench(B, H, W, cin, cout, mode=0, reps=5):
    x = torch.randn(B, H, W, cin, device=hu.DEV); w = torch.randn(cout, cin, 3, 3) * 0.05
    wd = hu.dev(w); wp = torch.empty(ctx.lib.nd_pack_conv3x3_wino_weight_floats(cin, cout), device=hu.DEV)
    L.call("nd_pack_conv3x3_wino_weight", wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream); ctx.sync()
    b = torch.randn(cout, device=hu.DEV); mad = torch.rand(B, 3, cin, device=hu.DEV) + 0.5
    out = torch.empty(B, H, W, cout, device=hu.DEV)
    slots = ctx.lib.nd_conv3x3_wino_stat_slots(H, W)
    st = torch.empty(B, slots, cout, 2, device=hu.DEV); sc = torch.empty(slots, device=hu.DEV)
    torch.cuda.synchronize()
    d = L.Conv3x3(); d.src, d.weight, d.bias, d.out = hu.src(x, None, mode, mad=mad), wp.data_ptr(), b.data_ptr(), out.data_ptr()
    d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
    d.stats, d.slot_count = st.data_ptr(), sc.data_ptr()
    e0, e1 = C.c_void_p(), C.c_void_p(); L.call("nd_event_create", C.byref(e0)); L.call("nd_event_create", C.byref(e1))
    L.call(ENTRY, C.byref(d), ctx.stream); ctx.sync()
    L.call("nd_event_record", e0, ctx.stream)
    for _ in range(reps): L.call(ENTRY, C.byref(d), ctx.stream)
    L.call("nd_event_record", e1, ctx.stream); ms = C.c_float(); L.call("nd_event_elapsed_ms", e0, e1, C.byref(ms))
    t = ms.value / reps
    return t * 1e3, 18.0 * cin * cout * H * W * B / t / 1e9
for sh in [(16, 256, 256, 64, 64), (16, 256, 256, 128, 64), (16, 64, 64, 256, 256), (16, 32, 32, 512, 512)]:
    us, tf = bench(*sh)
    print(ENTRY[12:18], os.environ.get("ND_LIB", "default")[-20:], sh, f"{us:8.1f} us {tf:6.1f} TF(alg)", flush=True)
