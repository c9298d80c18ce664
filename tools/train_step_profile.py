#!/usr/bin/env python3
"""One configuration of the training step (GaussianDiffusion.p_losses forward + backward + Adam on the NoiseDiffNet graph with the HIP operators)
for rocprofv3:  rocprofv3 --kernel-trace --stats -d gpurun_out/train_prof -- python3 tools/train_step_profile.py
Env: B (4), S (256), DIM (64), STEPS (6), HIP (1: .hip() operators, 0: PyTorch), NET (trainable | dropin: noisediff_amd.NoiseDiffNet under autograd),
ADAM (hip: noisediff_amd.train.Adam, one launch per step -- the default | torch: torch.optim.Adam's foreach form | fused: PyTorch's single-kernel Adam)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("ND_PKG_ROOT") or REPO)      # ND_PKG_ROOT: another copy of the package (A/B against a saved state)
from types import SimpleNamespace
import torch
from noisediff_amd import GaussianDiffusion, NoiseDiffNet, TrainableNoiseDiffNet, synth
torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda", 0)
B, S, STEPS, HIP = int(os.environ.get("B", 4)), int(os.environ.get("S", 256)), int(os.environ.get("STEPS", 6)), os.environ.get("HIP", "1") != "0"
DIM = int(os.environ.get("DIM", 64))
cond = {k: v.to(dev) for k, v in synth.make_condition(B, S, seed=1).items()}
img = synth.uniform(7, "img", (B, 4, S, S), -1.0, 1.0).to(dev)
torch.manual_seed(0)
if os.environ.get("NET", "trainable") == "dropin":
    net = NoiseDiffNet(SimpleNamespace(dim=DIM, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False)).to(dev).train()
else:
    net = TrainableNoiseDiffNet(SimpleNamespace(dim=DIM)).to(dev).hip(HIP)
gd = GaussianDiffusion(net, image_size=S, timesteps=1000, beta_schedule="sigmoid2", objective="pred_v").to(dev)
_adam = os.environ.get("ADAM", "hip")
if _adam == "hip":
    from noisediff_amd.train import Adam
    opt = Adam(net.parameters(), lr=1e-4)
else:
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, **({"fused": True} if _adam == "fused" else {}))   # (an explicit fused=False would also switch the foreach default off)


def one():
    opt.zero_grad(set_to_none=True)
    torch.manual_seed(1)
    loss = gd(img, cond)
    loss.backward()
    opt.step()
    return loss.detach()


for _ in range(3):
    one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS):
    loss = one()
torch.cuda.synchronize()
print(f"B={B} {S}x{S} dim={DIM} hip={HIP}: {(time.perf_counter() - t0) / STEPS * 1e3:.1f} ms/step, loss {float(loss):.6f}", flush=True)
if os.environ.get("PROFILE", "0") != "0":          # where do the ATen copies / adds / sums come from: per operator and input shape
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=True) as prof:
        for _ in range(2):
            one()
        torch.cuda.synchronize()
    ka = prof.key_averages(group_by_input_shape=True)
    if os.environ.get("PROFILE") == "2":                 # ATen / autograd-function rows only, per step
        rows = sorted(((e.self_device_time_total / 2, e.key, e.count // 2, str(e.input_shapes)[:110]) for e in ka if e.key.startswith(("aten::", "Optimizer"))
                       and e.self_device_time_total > 0), reverse=True)
        for us, key, cnt, shp in rows[:60]:
            print(f"{us:9.1f} us/step  {key:32s} x{cnt:4d}  {shp}")
    else:
        print(ka.table(sort_by="self_cuda_time_total", row_limit=70, max_name_column_width=48, max_shapes_column_width=90))
