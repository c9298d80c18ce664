#!/usr/bin/env python3
"""ND_PRO_TAIL (the previous ResnetBlock's tail applied on load by the next block's first conv, written through) against the two launches it replaces:
nd_affine_silu_add_f32 + nd_conv3x3_wino4_nhwc_f32 (plain source, statistics epilogue).  us per launch, HIP events on the library stream."""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
torch.zeros(1, device="cuda")
from noisediff_amd import _lib as L
import hiputil as hu
ctx = hu.Ctx()
REPS = int(os.environ.get("REPS", "10"))


def timed(fn):
    e0, e1 = C.c_void_p(), C.c_void_p(); L.call("nd_event_create", C.byref(e0)); L.call("nd_event_create", C.byref(e1))
    fn(); ctx.sync()
    L.call("nd_event_record", e0, ctx.stream)
    for _ in range(REPS): fn()
    L.call("nd_event_record", e1, ctx.stream); ms = C.c_float(); L.call("nd_event_elapsed_ms", e0, e1, C.byref(ms))
    return ms.value / REPS * 1e3


def case(B, H, W, c0, c1, cout):
    cin = c0 + c1
    t = torch.randn(B, H, W, c0, device=hu.DEV); r = torch.randn(B, H, W, c0, device=hu.DEV)
    sk = torch.randn(B, H, W, c1, device=hu.DEV) if c1 else None
    w = torch.randn(cout, cin, 3, 3) * 0.05
    wd = hu.dev(w); wp = torch.empty(ctx.lib.nd_pack_conv3x3_wino4_weight_floats(cin, cout), device=hu.DEV)
    L.call("nd_pack_conv3x3_wino4_weight", wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream); ctx.sync()
    b = torch.randn(cout, device=hu.DEV)
    mad = torch.rand(B, 3, c0, device=hu.DEV) + 0.5
    y = torch.empty(B, H, W, c0, device=hu.DEV); y2 = torch.empty(B, H, W, c0, device=hu.DEV)
    o1 = torch.empty(B, H, W, cout, device=hu.DEV); o2 = torch.empty(B, H, W, cout, device=hu.DEV)
    slots = ctx.lib.nd_conv3x3_wino4_stat_slots(H, W)
    st = torch.empty(B, slots, cout, 2, device=hu.DEV); sc = torch.empty(slots, device=hu.DEV)
    torch.cuda.synchronize()

    def desc(src, out):
        d = L.Conv3x3(); d.src, d.weight, d.bias, d.out = src, wp.data_ptr(), b.data_ptr(), out.data_ptr()
        d.stats, d.slot_count = st.data_ptr(), sc.data_ptr()
        d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
        return d
    d_plain = desc(hu.src(y, sk), o1)
    d_tail = desc(hu.src(t, sk, L.PRO_TAIL, mad=mad, map=r, gamma=y2), o2)
    tail = lambda: L.call("nd_affine_silu_add_f32", t.data_ptr(), c0, mad.data_ptr(), r.data_ptr(), c0, None, c0, y.data_ptr(), c0, B, H * W, c0, ctx.stream)
    u_tail = timed(tail)
    u_plain = timed(lambda: L.call("nd_conv3x3_wino4_nhwc_f32", C.byref(d_plain), ctx.stream))
    u_both = timed(lambda: (tail(), L.call("nd_conv3x3_wino4_nhwc_f32", C.byref(d_plain), ctx.stream)))
    u_fused = timed(lambda: L.call("nd_conv3x3_wino4_nhwc_f32", C.byref(d_tail), ctx.stream))
    ctx.sync()
    dy = (y - y2).abs().max().item(); do = (o1 - o2).abs().max().item() / o1.abs().max().item()
    print(f"({B},{H},{W}) {c0}+{c1}->{cout}: tail {u_tail:7.1f} + conv {u_plain:7.1f} = {u_tail + u_plain:7.1f} (back to back {u_both:7.1f}) | fused {u_fused:7.1f} us "
          f"| saves {u_both - u_fused:6.1f} us | y max diff {dy:.2e}, out rel diff {do:.2e}", flush=True)


B = int(os.environ.get("B", "16"))
for sh in [(B, 256, 256, 64, 0, 64), (B, 256, 256, 64, 64, 64), (B, 128, 128, 128, 0, 128), (B, 64, 64, 256, 0, 256), (B, 32, 32, 512, 0, 512), (B, 32, 32, 512, 512, 512)]:
    case(*sh)
