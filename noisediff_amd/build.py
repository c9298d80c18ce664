"""Build libnoisediff_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m noisediff_amd.build            # incremental
    python -m noisediff_amd.build --force

The library is the product: there is no fallback.  ``_lib.load()`` raises if it is missing.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libnoisediff_hip.so")
SOURCES = ["runtime", "conv3x3", "conv3x3_wino", "conv3x3_wino2", "conv3x3_wino4", "conv3x3_wgrad", "linear_wgrad", "pointwise", "pwchain", "norm", "norm_train", "small", "sampler",
           "attention", "linattn", "adam"]
ARCH = "gfx950"
# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs (gfx950 has a unified file); without it hipcc 7.2 parks
# them in AGPRs and wraps every v_mfma_f32_32x32x2_f32 in v_accvgpr_read/write copies (8 VALU per MFMA, measured)
BASE_FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
FLAGS = BASE_FLAGS + ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
# object name -> (source file, flags).  conv3x3_wino2 keeps its 256 accumulator registers in the AGPR half of the
# file on purpose (no VGPR-form switch) and its K loop must be unrolled completely (16 steps x 16 MFMA slices).
SPECIAL = {
    "conv3x3_wino2": ("conv3x3_wino2", BASE_FLAGS + ["-mllvm", "-pragma-unroll-threshold=1000000"]),
    # one wave per SIMD: 144 accumulator registers in the AGPR half, the rest of the pipeline state in the VGPR half
    "conv3x3_wino4": ("conv3x3_wino4", BASE_FLAGS),
}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libnoisediff_hip.so")
    return exe


def _local_includes(src: str, seen=None) -> list:
    """Files of csrc/ that `src` pulls in with #include "..." (recursively): a source that includes another SOURCE is stale when that one changes."""
    import re
    seen = set() if seen is None else seen
    out = []
    with open(src) as f:
        for name in re.findall(r'^\s*#\s*include\s+"([^"]+)"', f.read(), re.M):
            path = os.path.join(CSRC, name)
            if os.path.exists(path) and path not in seen:
                seen.add(path)
                out += [path] + _local_includes(path, seen)
    return out


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "nd_common.h"), os.path.join(HERE, "..", "include", "noisediff_hip.h")]
    jobs = []
    for name in SOURCES:
        srcname, flags = SPECIAL.get(name, (name, FLAGS))
        src, obj = os.path.join(CSRC, srcname + ".hip"), os.path.join(OBJ, name + ".o")
        if force or _stale(obj, [src] + headers + _local_includes(src)):
            jobs.append([hipcc, *flags, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        for err in ex.map(run, jobs):
            if err.strip() and verbose:
                print(err, file=sys.stderr)
    objs = [os.path.join(OBJ, n + ".o") for n in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
