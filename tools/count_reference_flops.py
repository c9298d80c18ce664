#!/usr/bin/env python3
"""ALGORITHMIC FLOPs of one NoiseDiffNet.forward per patch (SURVEY 8d's definition: 2 x MACs of every Conv2d / Linear the REFERENCE module runs,
counted with forward hooks) -- on the `meta` device, so nothing is computed.  Needs the reference tree (/root/reference: this container only; the two
unused imports torchvision / ema_pytorch are stubbed as in tests/golden/capture_golden.py).  Output at r4:
  64 256 -> 275.12 GFLOP (conv3x3 234.34)   64 128 -> 68.78   48 512 -> 627.61 (conv3x3 527.27: the reference's shipped workload, script.sh:10)   48 256 -> 156.91
bench.py's UNIT_GFLOP table holds these numbers."""
import sys, types, torch
sys.dont_write_bytecode = True
for m in ("torchvision", "torchvision.transforms", "torchvision.utils", "ema_pytorch"):
    sys.modules[m] = types.ModuleType(m)
sys.modules["ema_pytorch"].EMA = object
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]; sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
sys.modules["torchvision.transforms"].__dict__.update({"Compose": None})
sys.path.insert(0, "/root/reference")
from types import SimpleNamespace
from models.archs.Diffusion_arch import NoiseDiffNet
import torch.nn as nn
def count(dim, S):
    with torch.device("meta"):
        net = NoiseDiffNet(SimpleNamespace(dim=dim, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False))
    tot = {"conv3": 0, "conv1": 0, "conv7": 0, "linear": 0}
    def hook(m, i, o):
        if isinstance(m, nn.Conv2d):
            k = m.kernel_size[0]
            macs = o.numel() * m.in_channels * k * k
            tot["conv3" if k == 3 else "conv7" if k == 7 else "conv1"] += 2 * macs
        elif isinstance(m, nn.Linear):
            tot["linear"] += 2 * o.numel() * m.in_features
    for m in net.modules():
        if isinstance(m, (nn.Conv2d, nn.Linear)): m.register_forward_hook(hook)
    x = torch.empty(1, 4, S, S, device="meta"); t = torch.zeros(1, dtype=torch.long, device="meta")
    cond = {"clean_img": torch.empty(1, 4, S, S, device="meta"), "position": torch.empty(1, 2, S, S, device="meta"), "iso_ratio_idx": torch.zeros(1, dtype=torch.long, device="meta")}
    net(x, t, cond)
    return {k: v / 1e9 for k, v in tot.items()}, sum(tot.values()) / 1e9
for dim, S in ((64, 256), (64, 128), (48, 512), (48, 256)):
    print(dim, S, count(dim, S))
