import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
import hiputil as hu
from noisediff_amd import _lib as L
ctx = hu.Ctx()
B, HW, W = 16, 65536, 256
def bench(cin, cout, mode, act, res=False, vec=False, reps=5):
    x = torch.randn(B, HW, cin, device=hu.DEV); w = torch.randn(cout, cin) * 0.1
    wp = hu.pack_pw(ctx, w); b = torch.randn(cout, device=hu.DEV)
    g = torch.ones(cin, device=hu.DEV); be = torch.zeros(cin, device=hu.DEV); v = torch.randn(B, cin, device=hu.DEV)
    out = torch.empty(B, HW, cout, device=hu.DEV); r0 = torch.randn(B, HW, cout, device=hu.DEV) if res else None
    vo = torch.randn(B, cout, device=hu.DEV) if vec else None
    torch.cuda.synchronize()
    kw = dict(vec=v, gamma=g, beta=be) if mode == L.PRO_LAYERNORM else {}
    s = hu.src(x, None, mode, **kw)
    d = L.Pointwise(); d.src, d.weight, d.bias, d.out = s, wp.data_ptr(), b.data_ptr(), out.data_ptr()
    d.B, d.HW, d.W, d.cin, d.cout, d.ldo, d.act = B, HW, W, cin, cout, cout, act
    if res: d.res0, d.ldr0 = r0.data_ptr(), cout
    if vec: d.vec = vo.data_ptr()
    e0, e1 = C.c_void_p(), C.c_void_p(); L.call("nd_event_create", C.byref(e0)); L.call("nd_event_create", C.byref(e1))
    L.call("nd_pointwise_gemm_nhwc_f32", C.byref(d), ctx.stream); ctx.sync()
    L.call("nd_event_record", e0, ctx.stream)
    for _ in range(reps): L.call("nd_pointwise_gemm_nhwc_f32", C.byref(d), ctx.stream)
    L.call("nd_event_record", e1, ctx.stream); ms = C.c_float(); L.call("nd_event_elapsed_ms", e0, e1, C.byref(ms))
    return ms.value / reps * 1e3
for cin, cout in ((64, 64), (64, 128), (128, 64)):
    for mode, mname in ((L.PRO_NONE, "none"), (L.PRO_LAYERNORM, "LN"), (L.PRO_SILU, "silu")):
        for act, aname in ((0, "-"), (1, "gelu")):
            print(f"{cin}->{cout} pro={mname:5s} act={aname:5s} {bench(cin, cout, mode, act):7.1f} us   +res {bench(cin, cout, mode, act, res=True):7.1f}", flush=True)
