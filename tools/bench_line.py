#!/usr/bin/env python3
"""One-line summary of bench.py JSON lines read from the files named on the command line, or from stdin (ms/step, patches/s, conv3x3 family, per-kernel executed fractions) -- for A/B scripts."""
import fileinput, json
for l in fileinput.input():
    if not l.startswith("{"):
        continue
    j = json.loads(l)
    r = j.get("roofline") or {}
    ac = r.get("all_conv3x3") or {}
    ws = r.get("whole_step") or {}
    print(f"{j['ms_per_step']:.3f} ms/step  {j['value']:.4f} {j['unit']}  conv3x3 {ac.get('ms_per_step', 0):.3f} ms exec {ac.get('executed_frac', 0):.3f}  whole {ws.get('executed_frac', 0):.3f}  launches {ws.get('launches_per_step')}")
    for k, v in (r.get("by_kernel") or {}).items():
        print(f"    {k:44s} x{v['launches_per_step']:3d}  {v['avg_ms'] * 1e3:8.1f} us  exec {v['executed_frac']:.3f}")
