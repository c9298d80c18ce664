#!/usr/bin/env python3
"""Generate tests/golden/training.npz: the reference's q_sample / p_losses / forward on fixed inputs (build container only).

    python tests/golden/capture_training.py

The reference is imported read-only exactly as in capture_golden.py.  Inputs are noisediff_amd.synth streams
(weights seed 0, conditions seed 1, 'train.*' streams seed 5); the fixture holds outputs only: x_t, the pred_v target,
loss values for the three objectives (with and without offset noise, with per-sample timesteps), the loss of
GaussianDiffusion.forward with torch.randint / randn patched, and a few parameter gradients of the pred_v loss
(subsampled) that pin the backward pass a future HIP training path has to reproduce.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from capture_golden import import_reference, ref_net, sub, synth  # noqa: E402

DIM, B, H, T = 16, 2, 32, 1000
GRAD_KEYS = ["final_conv.weight", "downs.0.0.block1.proj.weight", "time_mlp.1.weight", "mid_block1.block2.norm.weight",
             "ups.3.2.ff.net.2.weight", "shot_mlp1.fc1.weight", "pos_block1.mlp.1.bias", "iso_embed.weight"]


def inputs():
    x0 = synth.uniform(5, "train.x0", (B, 4, H, H), -1.0, 1.0)
    noise = synth.make_noise(5, "train.noise", B, 4, H)
    t = torch.tensor([3, 777], dtype=torch.long)
    cond = synth.make_condition(B, H, seed=1)
    return x0, noise, t, cond


class Patched:
    """torch.randn -> the 'train.offset' stream; torch.randn_like -> 'train.noise'; torch.randint -> fixed timesteps."""

    def __enter__(self):
        self.saved = (torch.randn, torch.randn_like, torch.randint)
        torch.randn = lambda shape, *a, **k: synth.uniform(5, "train.offset", tuple(shape), -1.0, 1.0)
        torch.randn_like = lambda t_, *a, **k: synth.make_noise(5, "train.noise", B, 4, H)
        torch.randint = lambda lo, hi, shape, *a, **k: torch.tensor([3, 777], dtype=torch.long)
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like, torch.randint = self.saved


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ddp, arch = import_reference()
    out = {}
    x0, noise, t, cond = inputs()
    for objective in ("pred_v", "pred_noise", "pred_x0"):
        net = ref_net(arch, DIM).train()
        gd = ddp.GaussianDiffusion(torch.nn.DataParallel(net), image_size=H, timesteps=T, beta_schedule="sigmoid2", objective=objective)
        if objective == "pred_v":
            out["train.x_t"] = gd.q_sample(x0, t, noise.clone()).numpy()
            out["train.v_target"] = gd.predict_v(x0, t, noise).numpy()
        loss = gd.p_losses(x0, t, cond, noise=noise.clone())
        out[f"train.loss.{objective}"] = np.float64(loss.item())
        if objective == "pred_v":
            loss.backward()
            params = dict(net.named_parameters())
            for k in GRAD_KEYS:
                out[f"train.grad.{k}"] = sub(params[k].grad, 2048)
            out["train.grad_sq_norm"] = np.float64(sum(float((p.grad.double() ** 2).sum()) for p in params.values() if p.grad is not None))
            with Patched():
                out["train.loss.pred_v.offset0.1"] = np.float64(gd.p_losses(x0, t, cond, noise=noise.clone(), offset_noise_strength=0.1).item())
                out["train.loss.pred_v.forward"] = np.float64(gd(x0, cond).item())
    np.savez_compressed(os.path.join(HERE, "training.npz"), **out)
    for k, v in out.items():
        print(k, v if np.ndim(v) == 0 else np.shape(v))


if __name__ == "__main__":
    main()
