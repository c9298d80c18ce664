#!/usr/bin/env python3
"""Wave-state breakdown per kernel from one rocprofv3 --pmc pass (SQ counters, quad-cycles):
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU \\
            --output-format csv -d <dir> -- python3 bench.py --steps 2 --warmup 1 --soak-s 0 --no-cpu --no-roofline --eager
  python tools/pmc_wave_states.py <dir> [out.json]
Per kernel (summed over its launches): the share of wave time parked (s_waitcnt / barrier: WAIT_ANY), stalled at issue (WAIT_INST_ANY: dependencies, busy pipes)
and issuing (ACTIVE_INST_ANY), and inside the latter the VALU / LDS / VMEM shares (MI355X_MICROARCH.md: the three states are disjoint and add up to WAVE_CYCLES)."""
import collections, csv, glob, json, sys
f = glob.glob(f"{sys.argv[1]}/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        n[k] += 1
out = {}
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    w = c.get("SQ_WAVE_CYCLES", 0)
    if w <= 0 or k.startswith(("at::", "__amd")):
        continue
    row = {name: c.get(cn, 0) / w for name, cn in (("parked", "SQ_WAIT_ANY"), ("issue_stall", "SQ_WAIT_INST_ANY"), ("issuing", "SQ_ACTIVE_INST_ANY"),
                                                    ("issuing_valu", "SQ_ACTIVE_INST_VALU"), ("issuing_lds", "SQ_ACTIVE_INST_LDS"), ("issuing_vmem", "SQ_ACTIVE_INST_VMEM"))}
    row["valu_insts_per_launch"] = c.get("SQ_INSTS_VALU", 0) / max(n[k], 1)
    row["launches"] = n[k]
    out[k] = row
    print(f"{k[:44]:44s} x{n[k]:3d}  parked {row['parked']:.2f}  issue-stall {row['issue_stall']:.2f}  issuing {row['issuing']:.2f} (VALU {row['issuing_valu']:.2f} LDS {row['issuing_lds']:.2f} "
          f"VMEM {row['issuing_vmem']:.2f})  VALU insts/launch {row['valu_insts_per_launch']:.3g}")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
