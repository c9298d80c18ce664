#!/usr/bin/env python3
"""Experiment driver for the bf16 three-term F(4x4) kernel (tools/experiments/wino4b): ND_LIB=tools/_build/lib_w4b_<tag>.so python run.py [bench|clock|check]
  bench: us per launch of nd_conv3x3_wino4_nhwc_f32 (fp32 MFMA, product) and nd_conv3x3_wino4b_nhwc_f32 (experiment) on the bench layers, max difference
  clock: the -DW4_STAMP cycle split of the experiment (per region tile)
  check: both kernels against an fp64 convolution (max error / max |y|)"""
import os, sys, ctypes as C, statistics
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
import torch.nn.functional as F
torch.zeros(1, device="cuda")
from noisediff_amd import _lib as L
lib = L.load(os.environ["ND_LIB"])
import hiputil as hu
i32, i64, vp = C.c_int32, C.c_int64, C.c_void_p
lib.nd_conv3x3_wino4b_nhwc_f32.restype, lib.nd_conv3x3_wino4b_nhwc_f32.argtypes = i32, [C.POINTER(L.Conv3x3), vp]
lib.nd_pack_conv3x3_wino4b_weight.restype, lib.nd_pack_conv3x3_wino4b_weight.argtypes = i32, [vp, vp, i32, i32, vp]
lib.nd_pack_conv3x3_wino4b_weight_floats.restype, lib.nd_pack_conv3x3_wino4b_weight_floats.argtypes = i64, [i32, i32]
ctx = hu.Ctx()
mode = sys.argv[1] if len(sys.argv) > 1 else "bench"
SHAPES = [(16, 256, 256, 64, 64, 0), (16, 256, 256, 64, 64, 1), (16, 256, 256, 128, 64, 0), (16, 128, 128, 128, 128, 0), (16, 64, 64, 256, 256, 0),
          (16, 32, 32, 512, 512, 0), (16, 32, 32, 768, 512, 0)]
if os.environ.get("SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split("x")) for s in os.environ["SHAPES"].split(",")]


def setup(kind, B, H, W, cin, cout, mode_, stats=True):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, H, W, cin, generator=g); w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05; b = torch.randn(cout, generator=g)
    xd, wd, bd = hu.dev(x), hu.dev(w), hu.dev(b)
    if kind == "b":
        wp = torch.empty(lib.nd_pack_conv3x3_wino4b_weight_floats(cin, cout), device=hu.DEV)
        L.check(lib.nd_pack_conv3x3_wino4b_weight(wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream), "pack b")
    else:
        wp = torch.empty(lib.nd_pack_conv3x3_wino4_weight_floats(cin, cout), device=hu.DEV)
        L.call("nd_pack_conv3x3_wino4_weight", wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream)
    ctx.sync()
    out = torch.zeros(B, H, W, cout, device=hu.DEV)
    mad = hu.dev(torch.rand(B, 3, cin, generator=g) + 0.5)
    slots = lib.nd_conv3x3_wino4_stat_slots(H, W)
    st = torch.zeros(B, slots, cout, 2, device=hu.DEV); sc = torch.zeros(max(slots, 16 * 1024 * 2), device=hu.DEV)
    torch.cuda.synchronize()
    d = L.Conv3x3(); d.src, d.weight, d.bias, d.out = hu.src(xd, None, mode_, **({"mad": mad} if mode_ else {})), wp.data_ptr(), bd.data_ptr(), out.data_ptr()
    if stats:
        d.stats, d.slot_count = st.data_ptr(), sc.data_ptr()
    d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
    d._keep = (xd, wd, bd, wp, out, mad, st, sc)
    fn = lib.nd_conv3x3_wino4b_nhwc_f32 if kind == "b" else lib.nd_conv3x3_wino4_nhwc_f32
    return d, fn, out, (x, w, b, mad), sc


def time_us(d, fn, rounds=5, reps=8):
    e0, e1 = C.c_void_p(), C.c_void_p(); L.call("nd_event_create", C.byref(e0)); L.call("nd_event_create", C.byref(e1))
    for _ in range(3):
        L.check(fn(C.byref(d), ctx.stream), "launch")
    ctx.sync()
    ts = []
    for _ in range(rounds):
        L.call("nd_event_record", e0, ctx.stream)
        for _ in range(reps):
            fn(C.byref(d), ctx.stream)
        L.call("nd_event_record", e1, ctx.stream); ms = C.c_float(); L.call("nd_event_elapsed_ms", e0, e1, C.byref(ms))
        ts.append(ms.value / reps * 1e3)
    return statistics.median(ts)


if mode == "bench":
    tot = {"4": 0.0, "b": 0.0}
    for sh in SHAPES:
        res = {}
        for kind in ("4", "b"):
            d, fn, out, _, _ = setup(kind, *sh)
            res[kind] = (time_us(d, fn), out.clone())
            tot[kind] += res[kind][0]
        print(sh, f"fp32 {res['4'][0]:8.1f} us | bf16x3 {res['b'][0]:8.1f} us | ratio {res['b'][0] / res['4'][0]:.3f} | max diff {float((res['4'][1] - res['b'][1]).abs().max()):.2e}", flush=True)
    print("sum us:", {k: round(v, 1) for k, v in tot.items()}, "ratio", round(tot["b"] / tot["4"], 3))
elif mode == "check":
    for sh in SHAPES:
        B, H, W, cin, cout, m = sh
        Bc = min(B, 2)
        errs = {}
        for kind in ("4", "b"):
            d, fn, out, (x, w, b, mad), _ = setup(kind, Bc, H, W, cin, cout, m)
            L.check(fn(C.byref(d), ctx.stream), "launch"); ctx.sync()
            xin = x.permute(0, 3, 1, 2).double()
            if m:
                M_, A_, D_ = (mad.cpu().double()[:, i][:, :, None, None] for i in range(3))
                xin = F.silu((xin - M_) * A_ + D_)
            ref = F.conv2d(xin, w.double(), b.double(), padding=1)
            errs[kind] = float((out.permute(0, 3, 1, 2).cpu().double() - ref).abs().max() / ref.abs().max())
        print(sh, f"fp32 {errs['4']:.3e} | bf16x3 {errs['b']:.3e} | ratio {errs['b'] / errs['4']:.3f}", flush=True)
else:   # clock: needs a -DW4_STAMP build
    for (B, H, W, cin, cout, m) in SHAPES[:1] + SHAPES[5:6]:
        d, fn, out, _, sc = setup("b", B, H, W, cin, cout, m)
        dbg = torch.zeros(16 * 1024, dtype=torch.int64, device=hu.DEV)
        d.slot_count = dbg.data_ptr()
        torch.cuda.synchronize()
        for _ in range(3):
            L.check(fn(C.byref(d), ctx.stream), "launch"); ctx.sync()
        v = dbg.cpu().view(1024, 16).double()
        v = v[v[:, 2] > 0]
        cyc, real, chunks, epi, xf, second, first, last, third, wait, pro, top, stile = (v[:, i] for i in range(13))
        n_chunks = (cin + 15) // 16
        tiles = chunks / n_chunks
        pt = lambda t: float((t / tiles).mean())
        mhz = cyc / (real / 100.0)
        print((B, H, W, cin, cout), f"clock {mhz.mean():.0f} MHz; per tile {pt(cyc):.0f} cycles (MFMA pipe {3456 * n_chunks}) = stages of chunk 0 {pt(first):.0f}, 1 {pt(second):.0f}, 2 {pt(third):.0f}, last {pt(last):.0f};",
              f"all chunks {pt(cyc - epi - xf - wait - pro):.0f}; barrier waits {pt(wait):.0f}, transforms {pt(xf):.0f}, epilogue {pt(epi):.0f}; per WG stagger + prologue {float(pro.mean()):.0f}", flush=True)
