#!/usr/bin/env python3
"""nd_pointwise_gemm_nhwc_f32 and nd_pointwise_gemm_split_nhwc_f32 (bf16 x 3 split products) on the bench workload's wide 1x1 layers: error against fp64 and us / TF per layer.
ND_PW_BIG=0 (pipelined 64-pixel tiles) / 1 (large tiles, one wave per SIMD) / 2 (large tiles, 128 couts only): run once per value.
Errors are against an fp64 product, relative to the output's largest magnitude."""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch, torch.nn.functional as F
torch.zeros(1, device="cuda")
from noisediff_amd import _lib as L
if os.environ.get("ND_LIB"):
    L.load(os.environ["ND_LIB"])
import hiputil as hu
ctx = hu.Ctx()
B = 16
# (HW, cin, cout, first-source channels of a virtual concat, LayerNorm prologue + GELU, residual)
LAYERS = [(1024, 768, 512, 512, 0, 0), (1024, 512, 1024, 0, 1, 0), (1024, 1024, 512, 0, 0, 1), (1024, 512, 512, 0, 0, 1),
          (4096, 384, 256, 256, 0, 0), (4096, 256, 512, 0, 1, 0), (4096, 512, 256, 0, 0, 1), (4096, 256, 256, 0, 0, 1),
          (16384, 192, 128, 128, 0, 0), (16384, 128, 256, 0, 1, 0), (16384, 256, 128, 0, 0, 1), (16384, 128, 128, 0, 0, 1),
          (65536, 128, 128, 0, 0, 0)]
tot = {"fp32": 0.0, "split": 0.0}
rms_ratio = []


def pack_split(w):
    cout, cin = w.shape
    wd, out = hu.dev(w), torch.empty(ctx.lib.nd_pack_pointwise_weight_split_floats(cin, cout), device=hu.DEV)
    L.call("nd_pack_pointwise_weight_split", wd.data_ptr(), out.data_ptr(), cin, cout, ctx.stream); ctx.sync()
    return out
for (HW, cin, cout, c0, ln, res) in LAYERS:
    g = torch.Generator().manual_seed(HW + cin)
    x = torch.randn(B, HW, cin, generator=g); w = torch.randn(cout, cin, generator=g) / cin ** 0.5; b = torch.randn(cout, generator=g)
    xd, wp, bd = hu.dev(x), hu.pack_pw(ctx, w), hu.dev(b)
    wdev = hu.dev(w)
    x64, w64, b64 = xd.double(), wdev.double(), bd.double()          # fp64 reference on the device, from the operands the kernel reads
    r = torch.randn(B, HW, cout, generator=g) if res else None
    rd = hu.dev(r) if res else None
    keep = []
    if c0:
        xa, xb = hu.dev(x[..., :c0].contiguous()), hu.dev(x[..., c0:].contiguous()); s = hu.src(xa, xb); ref = F.linear(x64, w64, b64)
    elif ln:
        vec, gm, be = torch.randn(B, cin, generator=g), torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g)
        rs, vd = hu.full((B, HW, 2)), hu.dev(vec)
        L.call("nd_layernorm_stats_f32", xd.data_ptr(), cin, vd.data_ptr(), rs.data_ptr(), B, HW, cin, 1e-5, ctx.stream); ctx.sync()
        s = hu.src(xd, None, L.PRO_LAYERNORM, vec=vd, gamma=hu.dev(gm), beta=hu.dev(be), rowstats=rs)
        ref = F.gelu(F.linear(F.layer_norm(x64 + vd.double()[:, None], (cin,), hu.dev(gm).double(), hu.dev(be).double(), eps=1e-5), w64, b64))
    else:
        s = hu.src(xd); ref = F.linear(x64, w64, b64) + (rd.double() if res else 0)
    out = hu.full((B, HW, cout))
    d = L.Pointwise(); d.src, d.weight, d.bias, d.out = s, wp.data_ptr(), bd.data_ptr(), out.data_ptr()
    d.B, d.HW, d.W, d.cin, d.cout, d.ldo, d.act = B, HW, int(HW ** 0.5), cin, cout, cout, (L.ACT_GELU if ln else 0)
    if res: d.res0, d.ldr0 = rd.data_ptr(), cout
    cells, rms = [], {}
    for form, entry, wt in (("fp32", "nd_pointwise_gemm_nhwc_f32", wp), ("split", "nd_pointwise_gemm_split_nhwc_f32", pack_split(w))):
        d.weight = wt.data_ptr()
        out.zero_(); torch.cuda.synchronize()        # (zero_ runs on torch's stream, the kernel on ctx.stream)
        L.call(entry, C.byref(d), ctx.stream); ctx.sync()
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        rms[form] = float(((out.double() - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())
        e0, e1 = C.c_void_p(), C.c_void_p(); L.call("nd_event_create", C.byref(e0)); L.call("nd_event_create", C.byref(e1))
        reps = 10
        L.call("nd_event_record", e0, ctx.stream)
        for _ in range(reps): L.call(entry, C.byref(d), ctx.stream)
        L.call("nd_event_record", e1, ctx.stream); ms = C.c_float(); L.call("nd_event_elapsed_ms", e0, e1, C.byref(ms))
        us = ms.value / reps * 1e3; tot[form] += us
        ctx.sync()
        first, same = out.clone(), True                      # the timed launches wrote the same tensor ten times: it must still hold the first result, bit for bit
        for _ in range(3):
            L.call(entry, C.byref(d), ctx.stream); ctx.sync()
            same = same and torch.equal(out, first)
        err2 = float((out.double() - ref).abs().max() / ref.abs().max())
        cells.append(f"{form} {us:8.1f} us {2.0 * B * HW * cin * cout / us / 1e6:6.1f} TF max err {max(err, err2):.1e} rms {rms[form]:.1e}{'' if same else ' NOT REPEATABLE'}")
    rms_ratio.append(rms["split"] / rms["fp32"])
    print(f"{cin:5d} -> {cout:5d} @{HW:6d}px {'cat ' if c0 else 'LN+GELU ' if ln else 'res ' if res else ''}: " + " | ".join(cells), flush=True)
print("total us:", {k: round(v, 1) for k, v in tot.items()}, f"(ND_PW_BIG={os.environ.get('ND_PW_BIG', '1')})")
print(f"rms error of the split form / rms error of the fp32 form, against fp64: mean {sum(rms_ratio) / len(rms_ratio):.2f}, worst {max(rms_ratio):.2f}")
