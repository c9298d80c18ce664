#!/usr/bin/env python3
"""Golden vectors for the next row: LSID.forward of the REAL reference (build container only).
Writes tests/golden/lsid.npz; inputs and weights come from noisediff_amd.synth, outputs only are stored."""
import os, sys
from types import SimpleNamespace
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
from noisediff_amd import synth
from noisediff_amd.spec import lsid_param_spec
import models.archs.SID_arch as sid

net = sid.LSID(SimpleNamespace()).eval()
assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == [(p.name, p.shape) for p in lsid_param_spec()]
net.load_state_dict(synth.make_state_dict(lsid_param_spec(), 0), strict=True)
out = {}
with torch.no_grad():
    for (B, H, W) in ((2, 64, 64), (1, 36, 44)):      # the second size exercises ceil-mode pooling and the crop
        x = synth.uniform(9, f"lsid.x.{H}x{W}", (B, 4, H, W), 0.0, 1.0)
        out[f"lsid.{H}x{W}"] = net(x).numpy()
    # synth -> denoise composition + PSNR (BASELINE config 5 shape, scaled down)
    clean = synth.uniform(9, "lsid.clean", (2, 4, 64, 64), 0.0, 1.0)
    noise = synth.make_noise(9, "lsid.noise", 2, 4, 64) * 0.1
    noisy = np.clip(np.clip(noise.numpy(), -1.0, 1.0) + clean.numpy(), 0.0, 1.0)          # dataset_denoising.py:140-151
    den = net(torch.from_numpy(noisy)).clamp(0.0, 1.0)
    out["lsid.compose.out"] = den.numpy()
    out["lsid.compose.psnr"] = np.array(10.0 * np.log10(1.0 / np.mean((den.numpy().astype(np.float64) - clean.numpy()) ** 2)))
np.savez_compressed(os.path.join(HERE, "lsid.npz"), **out)
print({k: v.shape for k, v in out.items()}, float(out["lsid.compose.psnr"]))
