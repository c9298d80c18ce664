#!/usr/bin/env python3
"""Training slice: forward + backward of the bench workload's 3x3 conv layers, noisediff_amd.train.conv3x3 (HIP
library: forward and data gradient on the Winograd kernels, weight gradient on conv3x3_wgrad.hip) against PyTorch's own fp32
convolution autograd (MIOpen) on the same tensors.  ms per forward+backward, algorithmic TFLOP/s (3 x 18 Cin Cout per pixel)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch, torch.nn.functional as F
from noisediff_amd import train
torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda", 0)

def run(fn, x, w, b, gy, reps=5):
    xa, wa, ba = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    fn(xa, wa, ba).backward(gy)                       # warm-up (MIOpen picks its algorithms here)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        xa.grad = wa.grad = ba.grad = None
        fn(xa, wa, ba).backward(gy)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, (xa.grad, wa.grad)

for (B, H, W, cin, cout) in [(16, 256, 256, 64, 64), (16, 256, 256, 128, 64), (16, 64, 64, 256, 256), (16, 32, 32, 512, 512),
                           (4, 64, 64, 256, 256), (4, 32, 32, 512, 512), (4, 32, 32, 1024, 512)]:      # B = 4: the deep layers run on the split-K form
    x = torch.randn(B, cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (9 * cin) ** 0.5
    b = torch.randn(cout, device=dev)
    gy = torch.randn(B, cout, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    t_hip, g_hip = run(train.conv3x3, x, w, b, gy)
    t_ref, g_ref = run(lambda a, ww, bb: F.conv2d(a, ww, bb, padding=1), x, w, b, gy)
    err = max(float((p - q).abs().max() / q.abs().max()) for p, q in zip(g_hip, g_ref))
    fl = 3 * 18.0 * cin * cout * H * W * B
    print((B, H, W, cin, cout), f"HIP {t_hip * 1e3:7.2f} ms {fl / t_hip / 1e12:6.1f} TF | torch/MIOpen fp32 {t_ref * 1e3:7.2f} ms {fl / t_ref / 1e12:6.1f} TF | "
          f"x{t_ref / t_hip:.2f} | max rel grad diff {err:.1e}", flush=True)
