#!/usr/bin/env python3
"""us per call of nd_conv3x3_wgrad_nhwc_f32 (kernel + its reduce launches) on the 3x3 layers of the d=64 network's B=4 256x256 training step.
[ND_LIB=tools/_build/lib_<tag>.so] [ND_WGRAD_WINO=0] python tools/wgrad_bench.py     TF = 18 cin cout flops per pixel; GB/s = one read of x and dy."""
import os, sys, ctypes as C, statistics
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
torch.zeros(1, device="cuda")
from noisediff_amd import _lib as L
lib = L.load(os.environ["ND_LIB"]) if os.environ.get("ND_LIB") else L.load()
DEV = torch.device("cuda", 0)
SHAPES = [(4, 256, 256, 64, 64), (4, 256, 256, 128, 64), (4, 128, 128, 64, 64), (4, 128, 128, 128, 128), (4, 128, 128, 192, 128), (4, 64, 64, 128, 128),
          (4, 64, 64, 256, 256), (4, 64, 64, 384, 256), (4, 32, 32, 256, 256), (4, 32, 32, 512, 512), (4, 32, 32, 768, 512)]
if os.environ.get("SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split("x")) for s in os.environ["SHAPES"].split(",")]
st = torch.cuda.current_stream().cuda_stream
tot = 0.0
for (B, H, W, cin, cout) in SHAPES:
    x = torch.randn(B, H, W, cin, device=DEV); gy = torch.randn(B, H, W, cout, device=DEV)
    dw = torch.empty(cout, cin, 3, 3, device=DEV); db = torch.empty(cout, device=DEV)
    ws = torch.empty(int(lib.nd_conv3x3_wgrad_workspace_floats(B, H, W, cin, cout)), device=DEV)
    call = lambda: L.call("nd_conv3x3_wgrad_nhwc_f32", x.data_ptr(), cin, gy.data_ptr(), cout, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), B, H, W, cin, cout, st)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 100.0)
    us = statistics.median(ts); tot += us
    px = B * H * W
    print((B, H, W, cin, cout), f"{us:8.1f} us  {18.0 * cin * cout * px / us / 1e6:6.1f} TF  {4.0 * px * (cin + cout) / us / 1e3:7.0f} GB/s", flush=True)
print(f"sum {tot:.1f} us")
