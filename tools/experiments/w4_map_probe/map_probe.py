#!/usr/bin/env python3
"""How much of wino4_kernel<2>'s (per-pixel scale / shift map prologue, pos_block{1,2}.block2) time is the maps' HBM traffic?  Times the instance as it is and a
side build (ND_LIB) whose map loads all read the same 128 bytes per chunk (-DW4_MAP_SAME_PIXEL: same instructions, same waits, no map traffic) -- the upper bound of
what forming the maps inside the kernel from pos_emb could buy if its arithmetic were free.  Also the affine + SiLU instance without maps, for scale."""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
torch.zeros(1, device="cuda")
torch.manual_seed(0)
from noisediff_amd import _lib as L
if os.environ.get("ND_LIB"):
    L.load(os.environ["ND_LIB"])
import hiputil as hu
ctx = hu.Ctx()
REPS = 20
B, H, W, cin, cout = 16, 256, 256, 64, 64
x = torch.randn(B, H, W, cin, device=hu.DEV); w = torch.randn(cout, cin, 3, 3) * 0.05
wd = hu.dev(w); wp = torch.empty(ctx.lib.nd_pack_conv3x3_wino4_weight_floats(cin, cout), device=hu.DEV)
L.call("nd_pack_conv3x3_wino4_weight", wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream); ctx.sync()
b = torch.randn(cout, device=hu.DEV); out = torch.empty(B, H, W, cout, device=hu.DEV)
mad = torch.rand(B, 3, cin, device=hu.DEV) + 0.5
maps = torch.randn(B, H, W, 2 * cin, device=hu.DEV) * 0.1
pe = torch.randn(B, H, W, 8, device=hu.DEV); gw = torch.randn(2 * cin, 8, device=hu.DEV) * 0.1; gb = torch.randn(2 * cin, device=hu.DEV) * 0.1
slots = ctx.lib.nd_conv3x3_wino4_stat_slots(H, W)
st = torch.empty(B, slots, cout, 2, device=hu.DEV); sc = torch.empty(slots, device=hu.DEV)
torch.cuda.synchronize()
for name, src in (("plain", hu.src(x)), ("affine + SiLU", hu.src(x, None, L.PRO_AFFINE_SILU, mad=mad)),
                  ("affine + map + SiLU (blocked maps)", hu.src(x, None, L.PRO_AFFINE_MAP_SILU, mad=mad, map=maps, map_blocked=1)),
                  ("affine + maps formed in the kernel", hu.src(x, None, L.PRO_AFFINE_GENMAP_SILU, mad=mad, map=pe, gamma=gw, beta=gb))):
    d = L.Conv3x3(); d.src, d.weight, d.bias, d.out = src, wp.data_ptr(), b.data_ptr(), out.data_ptr()
    d.stats, d.slot_count = st.data_ptr(), sc.data_ptr()
    d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
    e0, e1 = C.c_void_p(), C.c_void_p(); L.call("nd_event_create", C.byref(e0)); L.call("nd_event_create", C.byref(e1))
    L.call("nd_conv3x3_wino4_nhwc_f32", C.byref(d), ctx.stream); ctx.sync()
    L.call("nd_event_record", e0, ctx.stream)
    for _ in range(REPS): L.call("nd_conv3x3_wino4_nhwc_f32", C.byref(d), ctx.stream)
    L.call("nd_event_record", e1, ctx.stream); ms = C.c_float(); L.call("nd_event_elapsed_ms", e0, e1, C.byref(ms))
    ctx.sync()
    print(f"{os.environ.get('ND_LIB', 'tree build'):40s} {name:36s} {ms.value / REPS * 1e3:7.1f} us   checksum {out.double().abs().sum().item():.10e} stats {st.double().abs().sum().item():.10e}", flush=True)
