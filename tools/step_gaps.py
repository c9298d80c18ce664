#!/usr/bin/env python3
"""Idle time between the kernels of a replayed diffusion step: python tools/step_gaps.py <..._kernel_trace.csv>
(from `rocprofv3 --kernel-trace --output-format csv -- python3 bench.py --steps 5 --warmup 1 --soak-s 0 --no-cpu --no-roofline`).
Takes the last 3 x 159-launch windows that start with cond_step_kernel; prints span, busy time, the gap histogram and the largest gaps."""
import csv, sys
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda e: e[0])
starts = [i for i, e in enumerate(ev) if "cond_step_kernel" in e[2]]
starts = [i for i in starts if i + 1 < len(ev)]
if len(starts) < 4:
    sys.exit("fewer than four steps in the trace")
a, b = starts[-4], starts[-1]
win = ev[a:b]
steps = 3
span = (win[-1][1] - win[0][0]) / 1e6 / steps
busy = sum(e - s for s, e, _ in win) / 1e6 / steps
gaps = [(win[i + 1][0] - win[i][1], win[i][2][:60], win[i + 1][2][:60]) for i in range(len(win) - 1)]
print(f"{len(win) / steps:.0f} launches per step; first start to last end {span:.3f} ms per step, kernels busy {busy:.3f} ms, idle between kernels {span - busy:.3f} ms "
      f"({(span - busy) / span * 100:.1f} %)")
gs = sorted(g[0] for g in gaps)
print("gap ns: median", gs[len(gs) // 2], "p90", gs[int(len(gs) * 0.9)], "max", gs[-1], "negative (overlap)", sum(1 for g in gs if g < 0))
for g in sorted(gaps, reverse=True)[:8]:
    print(f"  {g[0] / 1e3:7.1f} us between {g[1]} -> {g[2]}")
