#!/usr/bin/env python3
"""Layer-by-layer GPU-vs-oracle error report (run on the GPU box; writes gpurun_out/diag.txt)."""
import os
import sys
from types import SimpleNamespace

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from noisediff_amd import NoiseDiffNet, synth          # noqa: E402
from oracle import noisediff_oracle as O               # noqa: E402
from util import rel_err, state_dict                   # noqa: E402


def main():
    dim, B, H = int(os.environ.get("DIM", 16)), 2, int(os.environ.get("SIZE", 32))
    dev = torch.device("cuda", 0)
    sd = state_dict(dim)
    net = NoiseDiffNet(SimpleNamespace(dim=dim, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False))
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    cond = synth.make_condition(B, H, seed=1)
    x = synth.make_noise(2, "net.x", B, 4, H)
    t = torch.full((B,), 500, dtype=torch.long)
    taps = {}
    with torch.no_grad():
        ref = O.noisediff_forward(sd, x, t, cond, taps=taps)
    plan = net.hip_engine(dev).plan(B, H, H, debug=True)
    plan.set_condition({k: v.to(dev) for k, v in cond.items()})
    got = plan.forward(x.to(dev), t)
    lines = [f"final rel_err {rel_err(got.cpu().numpy(), ref.numpy()):.3e}"]
    for name, r in taps.items():
        g = plan.taps.get(name)
        if name == "pos_emb":
            g = plan.pos_emb
        if name == "t_emb" or g is None:
            continue
        gg = g.view(B, -1, g.shape[-1]).permute(0, 2, 1).reshape(r.shape).cpu()
        lines.append(f"{name:16s} rel_err {rel_err(gg.numpy(), r.numpy()):.3e}  nan={bool(torch.isnan(gg).any())}")
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "diag.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
