#!/usr/bin/env python3
"""Per-launch timing of one diffusion step (HIP events on the library stream, eager launches).
Writes gpurun_out/layer_times.txt: every op of the step with ms, algorithmic GB/s and TFLOP/s."""
import argparse, ctypes as C, os, sys
from types import SimpleNamespace
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from noisediff_amd import GaussianDiffusion, NoiseDiffNet, synth, _lib as L
from noisediff_amd.spec import noisediff_param_spec

ap = argparse.ArgumentParser()
ap.add_argument("--dim", type=int, default=64); ap.add_argument("--size", type=int, default=256)
ap.add_argument("--batch", type=int, default=16); ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda", 0)
net = NoiseDiffNet(SimpleNamespace(dim=a.dim, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False))
net.load_state_dict(synth.make_state_dict(noisediff_param_spec(a.dim), 0))
net = net.to(dev).eval()
plan = net.hip_engine(dev).plan(a.batch, a.size, a.size)
plan.set_condition({k: v.to(dev) for k, v in synth.make_condition(a.batch, a.size, seed=1).items()})
plan.load_x(synth.make_noise(2, "x_T", a.batch, 4, a.size))
st = plan.e.stream
ops = plan.step_ops
evs = []
for _ in range(len(ops) + 1):
    e = C.c_void_p(); L.call("nd_event_create", C.byref(e)); evs.append(e)
tot = [0.0] * len(ops)
for rep in range(a.reps + 1):
    L.call("nd_event_record", evs[0], st)
    for i, (fn, args, name, meta) in enumerate(ops):
        L.check(fn(*args), name)
        L.call("nd_event_record", evs[i + 1], st)
    L.call("nd_stream_sync", st)
    if rep == 0:
        continue
    for i in range(len(ops)):
        ms = C.c_float(); L.call("nd_event_elapsed_ms", evs[i], evs[i + 1], C.byref(ms)); tot[i] += ms.value
lines = []
sums = {}
for i, (fn, args, name, meta) in enumerate(ops):
    ms = tot[i] / a.reps
    kind = name.replace("nd_", "").replace("_f32", "").replace("_nhwc", "")
    desc, gbs, tf = "", 0.0, 0.0
    if meta and "H" in meta:
        kind = "conv3x3_wino" if meta["tiling"] == 9001 else "conv3x3_wino2" if meta["tiling"] == 9002 else kind
        desc_mode = meta.get("mode", 0)
        fl = 18.0 * meta["cin"] * meta["cout"] * meta["H"] * meta["W"] * meta["B"]
        by = 4.0 * meta["B"] * meta["H"] * meta["W"] * (meta["cin"] + meta["cout"])
        desc = f"{meta['layer']} {meta['cin']}->{meta['cout']} @{meta['H']}x{meta['W']} t{meta['tiling']} m{desc_mode}"
        gbs, tf = by / ms / 1e6, fl / ms / 1e9
    elif meta and "stream_bytes" in meta:
        desc, gbs = meta["layer"], meta["stream_bytes"] / ms / 1e6
    elif meta:
        fl = meta.get("flop_per_px", 2.0 * meta["cin"] * meta["cout"]) * meta["HW"] * meta["B"]
        by = 4.0 * meta["B"] * meta["HW"] * (meta["cin"] + meta["cout"])
        desc = f"{meta['layer']} {meta['cin']}->{meta['cout']} @{meta['HW']}px"
        gbs, tf = by / ms / 1e6, fl / ms / 1e9
    lines.append(f"{i:3d} {kind:22s} {ms*1e3:9.1f} us  {gbs:8.0f} GB/s(alg) {tf:7.1f} TF  {desc}")
    sums[kind] = sums.get(kind, 0.0) + ms
lines.append("")
for k, v in sorted(sums.items(), key=lambda kv: -kv[1]):
    lines.append(f"{k:24s} {v:8.3f} ms/step")
lines.append(f"{'TOTAL':24s} {sum(sums.values()):8.3f} ms/step (eager, with event overhead)")
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
open(os.path.join(REPO, "gpurun_out", "layer_times.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
