#!/usr/bin/env python3
"""BASELINE config 5 consumer: LSID denoiser forward on the HIP kernels at 512x512: ms per forward, TFLOP/s.
96.8 GFLOP per 512x512 frame.  usage: python tools/lsid_bench.py [--batch 4] [--size 512] [--reps 5]"""
import argparse, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from noisediff_amd import LSID, synth
from noisediff_amd.spec import lsid_param_spec

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4); ap.add_argument("--size", type=int, default=512); ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
net = LSID(None)
net.load_state_dict(synth.make_state_dict(lsid_param_spec(), 0))
net = net.to(dev).eval()
x = synth.uniform(3, "lsid.x", (a.batch, 4, a.size, a.size), 0.0, 1.0).to(dev)
with torch.inference_mode():
    y = net(x)                      # builds the plan, packs weights
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        y = net(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
gflop = 96.8 * (a.size / 512.0) ** 2 * a.batch
print(f"LSID forward B={a.batch} {a.size}x{a.size}: {dt * 1e3:.2f} ms  ({gflop / dt / 1e3:.1f} TFLOP/s algorithmic, {a.batch / dt:.1f} frames/s); "
      f"output finite: {bool(torch.isfinite(y).all())}")
