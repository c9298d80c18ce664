#!/usr/bin/env python3
"""Family table of a rocprofv3 --kernel-trace --stats run of tools/train_step_profile.py:
python tools/train_kernel_table.py <..._kernel_trace.csv> [steps]   (the last steps of the trace)   or   <..._kernel_stats.csv> [steps in the run]"""
import csv, sys


def family(n):
    if "linear_wgrad" in n: return "Linear / 1x1 weight gradient (HIP)"
    if "wgrad" in n: return "conv3x3 weight gradient (HIP)"
    if any(k in n for k in ("wino4_kernel", "wino2_kernel", "conv3x3_kernel", "wino_kernel", "w4_splitk_reduce")) and "pack" not in n: return "conv3x3 forward + data gradient (HIP)"
    if "pointwise" in n and "pack" not in n: return "Linear / 1x1 forward + data gradient (HIP pointwise)"
    if n.startswith("Cijk"): return "library GEMM (rocBLAS / hipBLASLt)"
    if any(k in n for k in ("gs_", "gn_", "ln_fwd", "ln_bwd", "ln_dparam", "token_sum", "modsilu", "affine3", "affine_silu_add")): return "GroupNorm / LayerNorm / modulation forward + backward, token sums (HIP)"
    if "pack_" in n: return "weight packing (HIP)"
    if "adam_kernel" in n: return "Adam (HIP, one launch)"
    if "multi_tensor" in n: return "Adam (ATen foreach)"
    if "CUDAFunctor_add" in n: return "ATen add"
    if "direct_copy" in n: return "ATen copy"
    if "reduce_kernel" in n: return "ATen reduce"
    if "Cat" in n: return "ATen cat"
    if "at::native" in n: return "ATen other"
    return "other"


def steps_from_trace(path, want):
    """The last `want` complete training steps of a ..._kernel_trace.csv: a step ends with the optimizer's kernels (adam_kernel, or ATen's multi_tensor_apply cluster).  (The first
    steps of a fresh box carry MIOpen's algorithm search for the 7x7 stem -- naive / CK candidates of hundreds of ms -- and stay out.)"""
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))), key=lambda e: e[0])
    adam = [i for i in range(len(ev)) if "multi_tensor_apply" in ev[i][2] or "adam_kernel" in ev[i][2]]
    ends = [i for i, j in zip(adam, adam[1:] + [len(ev) + 1000]) if j - i > 100]      # the optimizer's kernels come in one cluster per step
    ends = ends[-(want + 1):]
    return ev[ends[0] + 1: ends[-1] + 1], len(ends) - 1


if sys.argv[1].endswith("_kernel_trace.csv"):
    ev, steps = steps_from_trace(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    fam = {}
    for s0, e0, name in ev:
        f = fam.setdefault(family(name), [0.0, 0.0])
        f[0] += (e0 - s0) / 1e6 / steps
        f[1] += 1.0 / steps
    print(f"GPU busy {sum(v[0] for v in fam.values()):.2f} ms per step, {sum(v[1] for v in fam.values()):.0f} launches per step (last {steps} steps of the trace; "
          f"first to last kernel {(ev[-1][1] - ev[0][0]) / 1e6 / steps:.2f} ms per step)")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        print(f"  {k:55s} {v[0]:7.3f} ms  {v[1]:7.1f} launches")
    sys.exit(0)
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 9
fam = {}
for r in rows:
    f = fam.setdefault(family(r["Name"]), [0.0, 0.0])
    f[0] += float(r["TotalDurationNs"]) / 1e6 / steps
    f[1] += int(r["Calls"]) / steps
print(f"GPU busy {sum(v[0] for v in fam.values()):.2f} ms per step, {sum(v[1] for v in fam.values()):.0f} launches per step ({steps} steps in the trace)")
for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:55s} {v[0]:7.3f} ms  {v[1]:7.1f} launches")
