#!/usr/bin/env python3
"""nd_pointwise_chain_nhwc_f32 (fp32 MFMA) and nd_pointwise_chain_split_nhwc_f32 (bf16 x 3 split products) on the bench workload's fused chains:
error against an fp64 evaluation of the same layers, us per launch, bitwise repeatability."""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch, torch.nn.functional as F
torch.zeros(1, device="cuda")
from noisediff_amd import _lib as L
import hiputil as hu
ctx = hu.Ctx()
B = int(os.environ.get("B", 16))
# (HW, widths, AttnBlock tail (LayerNorm + residuals) or Mlp)
CASES = [(65536, [64, 128, 64, 64], True), (16384, [64, 128, 64, 64], True), (65536, [64, 64, 64], False), (65536, [64, 64, 4], False), (65536, [8, 64, 64], False),
         (65536, [48, 96, 48, 48], True)]
tot, ratios = {"fp32": 0.0, "split": 0.0}, []
for HW, widths, tail in CASES:
    n = len(widths) - 1
    g = torch.Generator().manual_seed(HW + widths[0])
    x = torch.randn(B, HW, widths[0], generator=g)
    ws = [torch.randn(widths[i + 1], widths[i], generator=g) / widths[i] ** 0.5 for i in range(n)]
    bs = [torch.randn(widths[i + 1], generator=g) * 0.1 for i in range(n)]
    xd = hu.dev(x)
    wd, bd = [hu.dev(w) for w in ws], [hu.dev(b) for b in bs]
    x64 = xd.double()
    if tail:
        Cc = widths[0]
        vec, gm, be = hu.dev(torch.randn(B, Cc, generator=g)), hu.dev(torch.rand(Cc, generator=g) + 0.5), hu.dev(torch.randn(Cc, generator=g))
        src = hu.src(xd, None, L.PRO_LAYERNORM, vec=vec, gamma=gm, beta=be)
        x1 = x64 + vec.double()[:, None]
        h = F.gelu(F.linear(F.layer_norm(x1, (Cc,), gm.double(), be.double(), eps=1e-5), wd[0].double(), bd[0].double()))
        ref = F.linear(F.linear(h, wd[1].double(), bd[1].double()) + x1, wd[2].double(), bd[2].double()) + x64
    else:
        src = hu.src(hu.dev(x[..., :4].contiguous()), hu.dev(x[..., 4:].contiguous())) if widths[0] == 8 else hu.src(xd)
        ref = F.linear(F.gelu(F.linear(x64, wd[0].double(), bd[0].double())), wd[1].double(), bd[1].double())
    out = hu.full((B, HW, widths[-1]))
    cells, rms = [], {}
    for form, entry, pack in (("fp32", "nd_pointwise_chain_nhwc_f32", "nd_pack_chain_weight"), ("split", "nd_pointwise_chain_split_nhwc_f32", "nd_pack_chain_weight_split")):
        d = L.Chain()
        keep = []
        for i in range(n):
            wp = torch.empty(getattr(ctx.lib, pack + "_floats")(widths[i], widths[i + 1], int(i == 0)), device=hu.DEV)
            L.call(pack, wd[i].data_ptr(), wp.data_ptr(), widths[i], widths[i + 1], int(i == 0), ctx.stream)
            keep.append(wp)
            d.st[i].weight, d.st[i].bias, d.st[i].cin, d.st[i].cout = wp.data_ptr(), bd[i].data_ptr(), widths[i], widths[i + 1]
        ctx.sync()
        d.src, d.out, d.n_stages, d.B, d.HW, d.ldo = src, out.data_ptr(), n, B, HW, widths[-1]
        d.st[0].act = L.ACT_GELU
        if tail:
            d.st[1].res, d.st[2].res = L.CHAIN_RES_INPUT, L.CHAIN_RES_INPUT_RAW
        out.zero_(); torch.cuda.synchronize()
        L.call(entry, C.byref(d), ctx.stream); ctx.sync()
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        rms[form] = float(((out.double() - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())
        first = out.clone()
        e0, e1 = C.c_void_p(), C.c_void_p(); L.call("nd_event_create", C.byref(e0)); L.call("nd_event_create", C.byref(e1))
        reps = 10
        L.call("nd_event_record", e0, ctx.stream)
        for _ in range(reps): L.call(entry, C.byref(d), ctx.stream)
        L.call("nd_event_record", e1, ctx.stream); ms = C.c_float(); L.call("nd_event_elapsed_ms", e0, e1, C.byref(ms))
        us = ms.value / reps * 1e3; tot[form] += us
        ctx.sync()
        same = torch.equal(out, first)
        flop = 2.0 * B * HW * sum(widths[i] * widths[i + 1] for i in range(n))
        gbs = 4.0 * B * HW * (widths[0] + widths[-1]) / us / 1e3
        cells.append(f"{form} {us:8.1f} us {flop / us / 1e6:6.1f} TF {gbs:6.0f} GB/s max err {err:.1e} rms {rms[form]:.1e}{'' if same else ' NOT REPEATABLE'}")
    ratios.append(rms["split"] / rms["fp32"])
    print(f"{'->'.join(map(str, widths)):>16s} @{HW:6d}px x {B} {'LN tail' if tail else 'mlp    '}: " + " | ".join(cells), flush=True)
print("total us:", {k: round(v, 1) for k, v in tot.items()})
print(f"rms error of the split form / rms error of the fp32 form, against fp64: mean {sum(ratios) / len(ratios):.2f}, worst {max(ratios):.2f}")
