#!/usr/bin/env python3
"""Training slice end to end: forward + backward + Adam step of a stack of ResnetBlock-shaped modules (Conv3x3 -> GroupNorm
-> SiLU -> Conv3x3 -> GroupNorm -> SiLU + shortcut: Diffusion_arch.py:146-170) at the bench workload's full resolution, plain PyTorch
(MIOpen / ATen) against the same modules with noisediff_amd.train.accelerate().  Prints ms per step and where PyTorch's time goes."""
import os, sys, time, copy
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch, torch.nn.functional as F
from torch import nn
from noisediff_amd import train
torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda", 0)


class Block(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.c1, self.n1 = nn.Conv2d(cin, cout, 3, padding=1), nn.GroupNorm(8, cout)
        self.c2, self.n2 = nn.Conv2d(cout, cout, 3, padding=1), nn.GroupNorm(8, cout)
        self.res = nn.Conv2d(cin, cout, 1) if cin != cout else nn.Identity()

    def forward(self, x):
        h = F.silu(self.n1(self.c1(x)))
        return F.silu(self.n2(self.c2(h))) + self.res(x)


def step_time(net, x, target, reps=5):
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    def one():
        opt.zero_grad(set_to_none=True)
        loss = F.mse_loss(net(x), target)
        loss.backward()
        opt.step()
        return loss
    one(); one(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): loss = one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, float(loss)


def real_net():
    """The whole NoiseDiffNet (d = 64) under GaussianDiffusion.p_losses: forward + backward + Adam, PyTorch vs .hip()."""
    from types import SimpleNamespace
    from noisediff_amd import GaussianDiffusion, TrainableNoiseDiffNet, synth
    for (B, S) in [(4, 256), (8, 128)]:
        cond = {k: v.to(dev) for k, v in synth.make_condition(B, S, seed=1).items()}
        img = synth.uniform(7, "img", (B, 4, S, S), -1.0, 1.0).to(dev)
        res = []
        for hip in (False, True):
            torch.manual_seed(0)
            net = TrainableNoiseDiffNet(SimpleNamespace(dim=64)).to(dev).hip(hip)
            gd = GaussianDiffusion(net, image_size=S, timesteps=1000, beta_schedule="sigmoid2", objective="pred_v").to(dev)
            opt = torch.optim.Adam(net.parameters(), lr=1e-4)
            def one():
                opt.zero_grad(set_to_none=True)
                torch.manual_seed(1)
                loss = gd(img, cond)
                loss.backward()
                opt.step()
                return loss.detach()
            for _ in range(4): one()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8): loss = one()
            torch.cuda.synchronize()
            res.append(((time.perf_counter() - t0) / 8, float(loss), torch.cuda.max_memory_allocated() / 2**30))
            del net, gd, opt
            torch.cuda.empty_cache()
        print(f"NoiseDiffNet d=64, B={B} {S}x{S}: PyTorch {res[0][0] * 1e3:7.1f} ms/step | .hip() {res[1][0] * 1e3:7.1f} ms/step | x{res[0][0] / res[1][0]:.2f} | "
              f"loss {res[0][1]:.6f} / {res[1][1]:.6f} | peak memory {res[1][2]:.1f} GiB", flush=True)


if os.environ.get("REAL_NET", "1") != "0":
    real_net()
for (B, S, C, nblk) in [(8, 256, 64, 4), (8, 64, 256, 4)]:
    torch.manual_seed(0)
    ref = nn.Sequential(nn.Conv2d(4, C, 1), *[Block(C, C) for _ in range(nblk)], nn.Conv2d(C, 4, 1)).to(dev).to(memory_format=torch.channels_last)
    hip = copy.deepcopy(ref); n = train.accelerate(hip)
    x = torch.randn(B, 4, S, S, device=dev).contiguous(memory_format=torch.channels_last); target = torch.randn(B, 4, S, S, device=dev)
    t_ref, l_ref = step_time(ref, x, target)
    t_hip, l_hip = step_time(hip, x, target)
    print(f"B={B} {S}x{S} C={C}, {nblk} blocks ({n} convs on HIP): PyTorch {t_ref * 1e3:7.2f} ms/step | accelerated {t_hip * 1e3:7.2f} ms/step | x{t_ref / t_hip:.2f} | loss {l_ref:.6f} / {l_hip:.6f}", flush=True)
    if os.environ.get("PROFILE", "1") != "0":
        from torch.profiler import profile, ProfilerActivity
        for name, net in (("pytorch", ref), ("accelerated", hip)):
            opt = torch.optim.Adam(net.parameters(), lr=1e-4)
            with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
                for _ in range(2):
                    opt.zero_grad(set_to_none=True); F.mse_loss(net(x), target).backward(); opt.step()
                torch.cuda.synchronize()
            print(f"--- {name}: top device-time ops"); print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=60))
