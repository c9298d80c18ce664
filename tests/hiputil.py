"""Helpers to call the C ABI directly from tests (torch only as a device-memory allocator)."""
import ctypes as C

import torch

from noisediff_amd import _lib as L

DEV = torch.device("cuda", 0)


class Ctx:
    def __init__(self):
        self.lib = L.load()
        self.stream = C.c_void_p()
        L.call("nd_stream_create", C.byref(self.stream))

    def sync(self):
        L.call("nd_stream_sync", self.stream)


def _settle(t):
    # torch's copies / fills run on torch's stream; the library runs on its own non-blocking stream
    torch.cuda.synchronize(DEV)
    return t


def dev(t):
    return _settle(t.to(DEV).contiguous())


def full(shape, value=float("nan")):
    return _settle(torch.full(shape, value, device=DEV))


def nhwc(t):           # NCHW cpu -> NHWC gpu
    return _settle(t.permute(0, 2, 3, 1).contiguous().to(DEV))


def nchw(t):           # NHWC gpu -> NCHW cpu
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def src(t, t2=None, mode=L.PRO_NONE, **kw):
    s = L.Src()
    s.p0, s.c0, s.ld0 = t.data_ptr(), t.shape[-1], t.shape[-1]
    if t2 is not None:
        s.p1, s.c1, s.ld1 = t2.data_ptr(), t2.shape[-1], t2.shape[-1]
    s.mode = mode
    for k, v in kw.items():
        setattr(s, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    # the struct only holds raw pointers: keep the tensors alive (a freed block would be recycled by
    # torch's caching allocator for the next allocation, e.g. the NaN-filled output buffer)
    s._refs = [t, t2] + list(kw.values())
    return s


def pack_conv3(ctx, w):
    cout, cin = w.shape[:2]
    n = ctx.lib.nd_pack_conv3x3_weight_floats(cin, cout)
    wd, out = dev(w), torch.empty(n, device=DEV)
    L.call("nd_pack_conv3x3_weight", wd.data_ptr(), out.data_ptr(), cin, cout, ctx.stream)
    ctx.sync()
    return out


def pack_pw(ctx, w, unshuffle_c=0):
    w = w.reshape(w.shape[0], -1)
    cout, cin = w.shape
    n = ctx.lib.nd_pack_pointwise_weight_floats(cin, cout)
    wd, out = dev(w), torch.empty(n, device=DEV)
    L.call("nd_pack_pointwise_weight", wd.data_ptr(), out.data_ptr(), cin, cout, unshuffle_c, ctx.stream)
    ctx.sync()
    return out


def conv3x3(ctx, s, wp, bias, B, H, W, cin, cout, stats=False):
    out = full((B, H, W, cout))
    d = L.Conv3x3()
    d.src, d.weight, d.bias, d.out = s, wp.data_ptr(), None if bias is None else bias.data_ptr(), out.data_ptr()
    d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
    st = sc = None
    slots = 0
    if stats:
        slots = ctx.lib.nd_conv3x3_stat_slots(H, W, cout, B)
        st = full((B, slots, cout, 2))
        sc = full((slots,))
        d.stats, d.slot_count = st.data_ptr(), sc.data_ptr()
    L.call("nd_conv3x3_nhwc_f32", C.byref(d), ctx.stream)
    ctx.sync()
    return out, st, sc, slots


def pointwise(ctx, s, wp, bias, B, HW, W, cin, cout, act=0, res0=None, res1=None, vec=None, gn_t=None, gn_mad=None, entry="nd_pointwise_gemm_nhwc_f32"):
    out = full((B, HW, cout))
    d = L.Pointwise()
    d.src, d.weight, d.bias, d.out = s, wp.data_ptr(), None if bias is None else bias.data_ptr(), out.data_ptr()
    d.B, d.HW, d.W, d.cin, d.cout, d.ldo, d.act = B, HW, W, cin, cout, cout, act
    if res0 is not None:
        d.res0, d.ldr0 = res0.data_ptr(), res0.shape[-1]
    if res1 is not None:
        d.res1, d.ldr1 = res1.data_ptr(), res1.shape[-1]
    if vec is not None:
        d.vec = vec.data_ptr()
    if gn_t is not None:
        d.gn_t, d.ldt, d.gn_mad = gn_t.data_ptr(), gn_t.shape[-1], gn_mad.data_ptr()
    L.call(entry, C.byref(d), ctx.stream)
    ctx.sync()
    return out


def gn_finalize(ctx, st, sc, slots, gamma, beta, ss, B, Cc, groups):
    mad = full((B, 3, Cc))
    L.call("nd_groupnorm_finalize_f32", st.data_ptr(), sc.data_ptr(), slots, gamma.data_ptr(), beta.data_ptr(),
           None if ss is None else ss.data_ptr(), 0 if ss is None else ss.shape[-1], mad.data_ptr(), B, Cc, groups, 1e-5, ctx.stream)
    ctx.sync()
    return mad
