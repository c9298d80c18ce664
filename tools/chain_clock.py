#!/usr/bin/env python3
"""Diagnostic: per-wave phase split of chain_kernel (needs a -DCHAIN_STAMP side build of pwchain.hip: tools/variant.sh pwchain stamp -DCHAIN_STAMP; ND_LIB).
FORM=split|fp32."""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch, numpy as np
torch.zeros(1, device="cuda")
from noisediff_amd import _lib as L
lib = L.load(os.environ["ND_LIB"])
raw = C.CDLL(os.environ["ND_LIB"])
import hiputil as hu
ctx = hu.Ctx()
B = 16
form = os.environ.get("FORM", "split")
pack, entry = (("nd_pack_chain_weight_split", "nd_pointwise_chain_split_nhwc_f32") if form == "split" else ("nd_pack_chain_weight", "nd_pointwise_chain_nhwc_f32"))
for HW, widths, tail in [(65536, [64, 128, 64, 64], True), (65536, [64, 64, 64], False)]:
    n = len(widths) - 1
    x = torch.randn(B, HW, widths[0], device=hu.DEV)
    out = torch.zeros(B, HW, widths[-1], device=hu.DEV)
    d = L.Chain(); keep = []
    for i in range(n):
        w = hu.dev(torch.randn(widths[i + 1], widths[i]) / widths[i] ** 0.5); bb = hu.dev(torch.randn(widths[i + 1]))
        wp = torch.empty(getattr(ctx.lib, pack + "_floats")(widths[i], widths[i + 1], int(i == 0)), device=hu.DEV)
        L.call(pack, w.data_ptr(), wp.data_ptr(), widths[i], widths[i + 1], int(i == 0), ctx.stream)
        keep += [w, bb, wp]
        d.st[i].weight, d.st[i].bias, d.st[i].cin, d.st[i].cout = wp.data_ptr(), bb.data_ptr(), widths[i], widths[i + 1]
    if tail:
        vec, gm, be = hu.dev(torch.randn(B, widths[0])), hu.dev(torch.rand(widths[0]) + 0.5), hu.dev(torch.randn(widths[0]))
        src = hu.src(x, None, L.PRO_LAYERNORM, vec=vec, gamma=gm, beta=be)
        d.st[1].res, d.st[2].res = L.CHAIN_RES_INPUT, L.CHAIN_RES_INPUT_RAW
    else:
        src = hu.src(x)
    d.src, d.out, d.n_stages, d.B, d.HW, d.ldo = src, out.data_ptr(), n, B, HW, widths[-1]
    d.st[0].act = L.ACT_GELU
    ctx.sync(); torch.cuda.synchronize()
    for _ in range(3):
        L.call(entry, C.byref(d), ctx.stream); ctx.sync()
    e0, e1 = C.c_void_p(), C.c_void_p(); L.call("nd_event_create", C.byref(e0)); L.call("nd_event_create", C.byref(e1))
    L.call("nd_event_record", e0, ctx.stream)
    for _ in range(int(os.environ.get("REPS", 10))): L.call(entry, C.byref(d), ctx.stream)
    L.call("nd_event_record", e1, ctx.stream); ctx.sync(); ms = C.c_float(); L.call("nd_event_elapsed_ms", e0, e1, C.byref(ms))
    reps = int(os.environ.get("REPS", 10))
    print(f"  {reps} launches back to back: {ms.value * 1000 / reps:.1f} us each")
    n_waves = 256 * 16
    buf = (C.c_ulonglong * (n_waves * 10))()
    assert raw.nd_chain_debug_read(buf, n_waves * 10) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n_waves, 10).astype(np.float64)
    a = a[a[:, 9] > 0]
    tiles = a[:, 9]
    per = (a[:, :7] / tiles[:, None]).mean(0)
    clock = (a[:, 7] / a[:, 8]).mean() * 100.0
    names = ["rows+prologue", "stage0", "res/act0", "stage1", "res/act1", "stage2", "res/act2+store"]
    print(f"{form} {'->'.join(map(str, widths))}: {len(a)} waves x {tiles.mean():.1f} tiles, clock {clock:.0f} MHz, wall per wave {a[:, 8].mean() / 100:.1f} us; cycles per tile and wave: "
          + ", ".join(f"{nm} {v:.0f}" for nm, v in zip(names, per)) + f"; sum {per.sum():.0f}", flush=True)
