#!/usr/bin/env python3
"""A/B of conv3x3_wino4 builds on one box: `python tools/w4_ab.py libA.so libB.so ...` runs every library in its own process (round-robin, `--rounds`
times) on the layer shapes that carry the bench workload -- plain / GroupNorm-affine + SiLU / + per-pixel map inputs, with the statistics epilogue --
and prints the median us per shape and library plus the ratio to the first library."""
import os, subprocess, sys, json, statistics
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHAPES = [  # (B, H, W, cin, cout, mode, stats)
    (16, 256, 256, 64, 64, 0, 1), (16, 256, 256, 64, 64, 1, 1), (16, 256, 256, 64, 64, 2, 1), (16, 256, 256, 128, 64, 0, 1),
    (16, 128, 128, 128, 128, 0, 1), (16, 128, 128, 128, 128, 1, 1), (16, 64, 64, 256, 256, 0, 1), (16, 32, 32, 512, 512, 0, 1), (16, 32, 32, 768, 512, 0, 1),
]


def worker(lib):
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    import ctypes as C
    import torch
    torch.zeros(1, device="cuda")
    from noisediff_amd import _lib as L
    L.load(lib)
    import hiputil as hu
    ctx = hu.Ctx()
    res = {}
    for (B, H, W, cin, cout, mode, stats) in SHAPES:
        g = torch.Generator().manual_seed(1)
        x = hu.dev(torch.randn(B, H, W, cin, generator=g)); w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
        wd = hu.dev(w); wp = torch.empty(ctx.lib.nd_pack_conv3x3_wino4_weight_floats(cin, cout), device=hu.DEV)
        L.call("nd_pack_conv3x3_wino4_weight", wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream); ctx.sync()
        b = hu.dev(torch.randn(cout, generator=g)); out = torch.empty(B, H, W, cout, device=hu.DEV)
        mad = hu.dev(torch.rand(B, 3, cin, generator=g) + 0.5)
        mp = hu.dev(torch.rand(B, H, W, 2 * cin, generator=g) - 0.5) if mode == 2 else None
        slots = ctx.lib.nd_conv3x3_wino4_stat_slots(H, W)
        st = torch.empty(B, slots, cout, 2, device=hu.DEV); sc = torch.empty(slots, device=hu.DEV)
        torch.cuda.synchronize()
        kw = {"mad": mad} if mode else {}
        if mode == 2:
            kw.update(map=mp, map_blocked=1)
        d = L.Conv3x3(); d.src, d.weight, d.bias, d.out = hu.src(x, None, mode, **kw), wp.data_ptr(), b.data_ptr(), out.data_ptr()
        if stats:
            d.stats, d.slot_count = st.data_ptr(), sc.data_ptr()
        d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
        e0, e1 = C.c_void_p(), C.c_void_p(); L.call("nd_event_create", C.byref(e0)); L.call("nd_event_create", C.byref(e1))
        for _ in range(3):
            L.call("nd_conv3x3_wino4_nhwc_f32", C.byref(d), ctx.stream)
        ctx.sync()
        reps = 10
        L.call("nd_event_record", e0, ctx.stream)
        for _ in range(reps):
            L.call("nd_conv3x3_wino4_nhwc_f32", C.byref(d), ctx.stream)
        L.call("nd_event_record", e1, ctx.stream); ctx.sync()
        ms = C.c_float(); L.call("nd_event_elapsed_ms", e0, e1, C.byref(ms))
        res[str((B, H, W, cin, cout, mode, stats))] = [ms.value / reps * 1e3, float(out.double().abs().sum()), float(st.double().abs().sum())]
    print("RESULT " + json.dumps(res), flush=True)


def main():
    if sys.argv[1] == "--worker":
        return worker(sys.argv[2])
    rounds = 3
    libs = [a for a in sys.argv[1:] if not a.startswith("--")]
    for a in sys.argv[1:]:
        if a.startswith("--rounds="):
            rounds = int(a.split("=")[1])
    acc = {lib: {} for lib in libs}
    sums = {lib: {} for lib in libs}
    for _ in range(rounds):
        for lib in libs:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", os.path.abspath(lib)], capture_output=True, text=True, timeout=300)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
            if not line:
                print(f"{lib}: FAILED\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}")
                continue
            for k, (us, cs, ss) in json.loads(line[0][7:]).items():
                acc[lib].setdefault(k, []).append(us)
                sums[lib][k] = (cs, ss)
    print("shape (B, H, W, cin, cout, mode, stats) | " + " | ".join(os.path.basename(l) for l in libs))
    tot = {lib: 0.0 for lib in libs}
    for k in acc[libs[0]]:
        base = statistics.median(acc[libs[0]][k])
        cells = []
        for lib in libs:
            if k not in acc[lib]:
                cells.append("   -   ")
                continue
            m = statistics.median(acc[lib][k])
            tot[lib] += m
            same = "" if sums[lib][k] == sums[libs[0]][k] else " (sums differ)"
            cells.append(f"{m:7.1f} us x{base / m:.3f}{same}")
        print(f"{k:38s} | " + " | ".join(cells))
    print("sum".ljust(38) + " | " + " | ".join(f"{tot[l]:7.1f} us x{tot[libs[0]] / tot[l]:.3f}" if tot[l] else "-" for l in libs))


if __name__ == "__main__":
    main()
