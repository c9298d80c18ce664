#!/usr/bin/env python3
"""Training step of TrainableNoiseDiffNet(...).hip() as ONE captured graph (torch.cuda.CUDAGraph): forward + backward + Adam(capturable=True) --
noisediff_amd.train.Adam (one launch + a counter launch; env ADAM=torch: torch.optim.Adam).
The library's launches are queued on torch's current stream, so they are captured like any ATen kernel; replaying the graph removes the host side of the
~300 autograd-function calls of a step (which bounds the small configurations in eager mode)."""
import os, sys, time
sys.path.insert(0, os.environ.get("ND_PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # ND_PKG_ROOT: another copy of the package (A/B against a saved state)
import torch
from types import SimpleNamespace
from noisediff_amd import GaussianDiffusion, TrainableNoiseDiffNet, synth
from noisediff_amd.train import Adam as HipAdam
dev = torch.device("cuda", 0)
for (B, S) in [(4, 256), (8, 128)]:
    cond = {k: v.to(dev) for k, v in synth.make_condition(B, S, seed=1).items()}
    img = synth.uniform(7, "img", (B, 4, S, S), -1.0, 1.0).to(dev)
    net = TrainableNoiseDiffNet(SimpleNamespace(dim=64)).to(dev).hip(True)
    gd = GaussianDiffusion(net, image_size=S, timesteps=1000, beta_schedule="sigmoid2", objective="pred_v").to(dev)
    opt = (torch.optim.Adam if os.environ.get("ADAM") == "torch" else HipAdam)(net.parameters(), lr=1e-4, capturable=True)
    def one():
        opt.zero_grad(set_to_none=True)
        loss = gd(img, cond)
        loss.backward()
        opt.step()
        return loss.detach()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): one()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): l = one()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 5
    g = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        loss = gd(img, cond)
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8): g.replay()
    torch.cuda.synchronize()
    graphed = (time.perf_counter() - t0) / 8
    print(f"B={B} {S}x{S}: eager {eager * 1e3:.1f} ms/step, one captured graph per step {graphed * 1e3:.1f} ms/step, loss {float(loss.detach()):.6f}", flush=True)
