#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) into profiles/<tag>_traffic.json.

usage: tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> "<command that was profiled>"
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB and on gfx950 FETCH_SIZE
reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section)."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_blobs import source_blobs

def avg(dirname, counter):
    f = glob.glob(f"{dirname}/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}

fetch, write = avg(sys.argv[1], "FETCH_SIZE"), avg(sys.argv[2], "WRITE_SIZE")
out = {"command": sys.argv[4], "source_blobs": source_blobs(), "units": "KiB per launch (rocprofv3 raw); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024", "kernels": {}}
for k in sorted(fetch):
    name = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
    if name.startswith(("at::", "__amd")):
        continue
    f, n = fetch[k]
    w = write.get(k, (0.0, 0))[0]
    out["kernels"][name] = {"launches": n, "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
                            "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
