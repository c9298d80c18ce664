#!/usr/bin/env python3
"""conv3x3_wino4 on 16 x 16-pixel regions (two workgroups per CU, nd_conv3x3_wino4_16_nhwc_f32) against the one-workgroup form and wino2 on the layer
shapes of the bench workloads (cfg3: d=64 at 256x256; cfg2: d=64 at 128x128), plain and GroupNorm-affine + SiLU inputs with the statistics epilogue:
us per launch (median of `ROUNDS` timed groups), executed fraction of the fp32 matrix pipe for the F(4x4) forms, bit equality of the two F(4x4) forms.
ND_LIB=<side build> selects another library."""
import os, sys, ctypes as C, statistics
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
torch.zeros(1, device="cuda")
from noisediff_amd import _lib as L
if os.environ.get("ND_LIB"):
    L.load(os.environ["ND_LIB"])
import hiputil as hu
ctx = hu.Ctx()
ROUNDS = int(os.environ.get("ROUNDS", "5"))
ENTRIES = [e for e in os.environ.get("ENTRIES", "wino2,wino4,wino4_16").split(",")]
NAMES = {"wino2": ("nd_conv3x3_wino2_nhwc_f32", "nd_pack_conv3x3_wino_weight"), "wino4": ("nd_conv3x3_wino4_nhwc_f32", "nd_pack_conv3x3_wino4_weight"),
         "wino4_16": ("nd_conv3x3_wino4_16_nhwc_f32", "nd_pack_conv3x3_wino4_weight")}

SHAPES = [  # (B, H, W, cin, cout, mode)
    (16, 256, 256, 64, 64, 0), (16, 256, 256, 64, 64, 1), (16, 256, 256, 128, 64, 0), (16, 128, 128, 128, 128, 0), (16, 128, 128, 128, 128, 1),
    (16, 128, 128, 192, 128, 0), (16, 64, 64, 256, 256, 0), (16, 64, 64, 384, 256, 0), (16, 32, 32, 256, 256, 0), (16, 32, 32, 512, 512, 0), (16, 32, 32, 768, 512, 0),
    # cfg2 (128 x 128 patches)
    (16, 128, 128, 64, 64, 0), (16, 64, 64, 128, 128, 0), (16, 32, 32, 384, 256, 0), (16, 16, 16, 256, 256, 0), (16, 16, 16, 512, 512, 0), (16, 16, 16, 768, 512, 0),
]
if os.environ.get("SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split("x")) for s in os.environ["SHAPES"].split(",")]


def bench(kind, B, H, W, cin, cout, mode):
    entry, pack = NAMES[kind]
    g = torch.Generator().manual_seed(1)
    x = hu.dev(torch.randn(B, H, W, cin, generator=g)); w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    wd = hu.dev(w); wp = torch.empty(getattr(ctx.lib, pack + "_floats")(cin, cout), device=hu.DEV)
    L.call(pack, wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream); ctx.sync()
    b = hu.dev(torch.randn(cout, generator=g))
    out = torch.zeros(B, H, W, cout, device=hu.DEV)
    mad = hu.dev(torch.rand(B, 3, cin, generator=g) + 0.5)
    slots = ctx.lib.nd_conv3x3_wino4_stat_slots(H, W) if kind != "wino2" else ctx.lib.nd_conv3x3_wino_stat_slots(H, W)
    st = torch.zeros(B, slots, cout, 2, device=hu.DEV); sc = torch.zeros(slots, device=hu.DEV)
    torch.cuda.synchronize()
    d = L.Conv3x3(); d.src, d.weight, d.bias, d.out = hu.src(x, None, mode, **({"mad": mad} if mode else {})), wp.data_ptr(), b.data_ptr(), out.data_ptr()
    d.stats, d.slot_count = st.data_ptr(), sc.data_ptr()
    d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
    e0, e1 = C.c_void_p(), C.c_void_p(); L.call("nd_event_create", C.byref(e0)); L.call("nd_event_create", C.byref(e1))
    for _ in range(3):
        L.call(entry, C.byref(d), ctx.stream)
    ctx.sync()
    ts = []
    reps = 8
    for _ in range(ROUNDS):
        L.call("nd_event_record", e0, ctx.stream)
        for _ in range(reps):
            L.call(entry, C.byref(d), ctx.stream)
        L.call("nd_event_record", e1, ctx.stream); ms = C.c_float(); L.call("nd_event_elapsed_ms", e0, e1, C.byref(ms))
        ts.append(ms.value / reps * 1e3)
    return statistics.median(ts), out, st


tot = {k: 0.0 for k in ENTRIES}
for sh in SHAPES:
    B, H, W, cin, cout, mode = sh
    cells, outs = [], {}
    for kind in ENTRIES:
        us, out, st = bench(kind, *sh)
        tot[kind] += us
        outs[kind] = (out, st)
        frac = 18.0 * cin * cout * H * W * B / (us * 1e-6) / (4.0 if kind != "wino2" else 2.25) / 157.3e12
        cells.append(f"{kind} {us:8.1f} us ({frac:.3f})")
    same = ""
    if "wino4" in outs:
        for other in ("wino4_16",):
            if other in outs:
                same += f" | {other} bits " + ("EQUAL" if torch.equal(outs["wino4"][0], outs[other][0]) and torch.equal(outs["wino4"][1], outs[other][1]) else "DIFFER")
    print(sh, " | ".join(cells) + same, flush=True)
print("sum us:", {k: round(v, 1) for k, v in tot.items()})
