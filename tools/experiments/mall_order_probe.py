#!/usr/bin/env python3
"""Does a consumer find its producer's output in the L2s / Infinity Cache when it reads it in the OPPOSITE order?  A 268 MB tensor (16 x 256 x 256 x 64 fp32) is written by
nd_affine_silu_add_f32 (ordinary stores, samples 0 .. 15) and read by the 64 -> 64 F(4x4) convolution as two launches of 8 samples: first half first (the order of today: the
cache holds the END of the tensor when the consumer starts at its beginning) against second half first.  us per (producer + both consumer launches), HIP events."""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
torch.zeros(1, device="cuda")
from noisediff_amd import _lib as L
import hiputil as hu
ctx = hu.Ctx()
REPS = 10
B, H, W, cin, cout = 16, 256, 256, 64, 64
t = torch.randn(B, H, W, cin, device=hu.DEV); r = torch.randn(B, H, W, cin, device=hu.DEV); y = torch.empty(B, H, W, cin, device=hu.DEV)
mad = torch.rand(B, 3, cin, device=hu.DEV) + 0.5
w = torch.randn(cout, cin, 3, 3) * 0.05
wd = hu.dev(w); wp = torch.empty(ctx.lib.nd_pack_conv3x3_wino4_weight_floats(cin, cout), device=hu.DEV)
L.call("nd_pack_conv3x3_wino4_weight", wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream); ctx.sync()
b = torch.randn(cout, device=hu.DEV); out = torch.empty(B, H, W, cout, device=hu.DEV)
slots = ctx.lib.nd_conv3x3_wino4_stat_slots(H, W)
st = torch.empty(B, slots, cout, 2, device=hu.DEV); sc = torch.empty(slots, device=hu.DEV)
torch.cuda.synchronize()


def piece(b0, nb):
    d = L.Conv3x3(); s = hu.src(y[b0:b0 + nb])
    d.src, d.weight, d.bias, d.out = s, wp.data_ptr(), b.data_ptr(), out[b0:b0 + nb].data_ptr()
    d.stats, d.slot_count = st[b0:b0 + nb].data_ptr(), sc.data_ptr()
    d.B, d.H, d.W, d.cin, d.cout, d.ldo = nb, H, W, cin, cout, cout
    return d


tail = lambda: L.call("nd_affine_silu_add_f32", t.data_ptr(), cin, mad.data_ptr(), r.data_ptr(), cin, None, cin, y.data_ptr(), cin, B, H * W, cin, ctx.stream)
whole, lo, hi = piece(0, B), piece(0, B // 2), piece(B // 2, B // 2)
q = [piece(i * 4, 4) for i in range(4)]
e8 = [piece(i * 2, 2) for i in range(8)]
conv = lambda d: L.call("nd_conv3x3_wino4_nhwc_f32", C.byref(d), ctx.stream)


def timed(fn):
    e0, e1 = C.c_void_p(), C.c_void_p(); L.call("nd_event_create", C.byref(e0)); L.call("nd_event_create", C.byref(e1))
    fn(); ctx.sync()
    L.call("nd_event_record", e0, ctx.stream)
    for _ in range(REPS): fn()
    L.call("nd_event_record", e1, ctx.stream); ms = C.c_float(); L.call("nd_event_elapsed_ms", e0, e1, C.byref(ms))
    return ms.value / REPS * 1e3


for name, fn in (("consumer alone, one launch of 16", lambda: conv(whole)), ("consumer alone, halves", lambda: (conv(lo), conv(hi))),
                 ("consumer alone, quarters", lambda: (conv(q[0]), conv(q[1]), conv(q[2]), conv(q[3]))),
                 ("consumer alone, eighths", lambda: [conv(e) for e in e8]),
                 ("producer alone", tail), ("producer + consumer, one launch of 16", lambda: (tail(), conv(whole))),
                 ("producer + consumer halves in the producer's order", lambda: (tail(), conv(lo), conv(hi))),
                 ("producer + consumer halves in the opposite order", lambda: (tail(), conv(hi), conv(lo))),
                 ("producer + consumer quarters in the producer's order", lambda: (tail(), conv(q[0]), conv(q[1]), conv(q[2]), conv(q[3]))),
                 ("producer + consumer quarters in the opposite order", lambda: (tail(), conv(q[3]), conv(q[2]), conv(q[1]), conv(q[0])))):
    print(f"{name:60s} {timed(fn):8.1f} us", flush=True)
