#!/usr/bin/env python3
"""Turn one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass into profiles/<tag>_mfma_busy.json.

usage: tools/pmc_mfma.py <pmc_dir> <out.json> "<command that was profiled>"
matrix-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (n_SIMD * GRBM_GUI_ACTIVE / 8): the SQ counter is summed over every
SIMD of the chip (256 CUs x 4), GRBM_GUI_ACTIVE over the 8 XCDs (MI355X_MICROARCH.md, DVFS note)."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_blobs import source_blobs

f = glob.glob(f"{sys.argv[1]}/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"command": sys.argv[3], "source_blobs": source_blobs(), "n_simd": 1024, "kernels": {}}
for k, c in sorted(agg.items()):
    name = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
    if name.startswith(("at::", "__amd")) or "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
        continue
    busy, gui = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]), sum(c["GRBM_GUI_ACTIVE"])
    if busy <= 0 or gui <= 0:
        continue
    out["kernels"][name] = {"launches": len(c["GRBM_GUI_ACTIVE"]), "mfma_busy_cycles_per_launch": busy / len(c["GRBM_GUI_ACTIVE"]),
                            "gui_active_per_xcd_per_launch": gui / 8 / len(c["GRBM_GUI_ACTIVE"]),
                            "matrix_pipe_busy_frac": busy / (1024 * gui / 8)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out["kernels"].items():
    print(f"{k:50s} busy {v['matrix_pipe_busy_frac']:.3f}  ({v['launches']} launches)")
