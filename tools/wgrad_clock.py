#!/usr/bin/env python3
"""Cycle split of the Winograd-domain weight-gradient kernel (conv3x3_wgrad.hip built with -DWW_STAMP: tools/variant.sh conv3x3_wgrad ww_STAMP -DWW_STAMP):
ND_LIB=tools/_build/lib_ww_STAMP.so python tools/wgrad_clock.py     per wave role: cycles per pipeline iteration = wait for the first operands /
the 18 MFMA slots / wait at the barrier; the shader clock under load; the share of the kernel outside the steady loop."""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
torch.zeros(1, device="cuda")
from noisediff_amd import _lib as L
lib = L.load(os.environ["ND_LIB"])
DEV = torch.device("cuda", 0)
SHAPES = [(4, 256, 256, 64, 64), (4, 128, 128, 128, 128), (4, 32, 32, 512, 512)]
if os.environ.get("SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split("x")) for s in os.environ["SHAPES"].split(",")]
st = torch.cuda.current_stream().cuda_stream
for (B, H, W, cin, cout) in SHAPES:
    x = torch.randn(B, H, W, cin, device=DEV); gy = torch.randn(B, H, W, cout, device=DEV)
    dw = torch.empty(cout, cin, 3, 3, device=DEV); db = torch.empty(cout, device=DEV)
    n_ws = int(lib.nd_conv3x3_wgrad_workspace_floats(B, H, W, cin, cout))
    n_wg = max(256, (cin // 32) * (cout // 32))
    ws = torch.zeros(n_ws + n_wg * 4 * 8 + 1024, device=DEV)
    for _ in range(3):
        L.call("nd_conv3x3_wgrad_nhwc_f32", x.data_ptr(), cin, gy.data_ptr(), cout, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), B, H, W, cin, cout, st)
    torch.cuda.synchronize()
    blocks = (cin // 32) * (cout // 32)
    groups = B * (H // 4) * (W // 16)
    S = min((-(-blocks * c // 256) * (-(-groups // c) + 14), c) for c in range(1, min(groups, 1024) + 1) if -(-blocks * c // 256) <= 9)[1]      # ww_plan
    base = S * 36 * cout * cin + S * cout
    d = ws[base: base + blocks * S * 32].cpu().view(-1, 4, 8).double()
    cyc, real, n, a, b, c = (d[:, :, i] for i in range(6))
    mhz = (cyc / (real / 100.0)).mean()
    print((B, H, W, cin, cout), f"S={S} workgroups={blocks * S} iterations/WG={float(n.mean()):.1f} clock {mhz:.0f} MHz kernel {float(cyc.mean()):.0f} cycles, "
          f"steady loop {float(((a + b + c).sum(1) / 4).mean()):.0f}")
    for w, name in enumerate(("V rows 0-2", "V rows 3-5", "D rows 0-2 (+bias)", "D rows 3-5")):
        print(f"   wave {w} {name:20s} per iteration: first operands {float((a[:, w] / n[:, w]).mean()):6.0f}  slots {float((b[:, w] / n[:, w]).mean()):6.0f}  "
              f"barrier {float((c[:, w] / n[:, w]).mean()):6.0f}  total {float(((a + b + c)[:, w] / n[:, w]).mean()):6.0f}   (MFMA pipe 1152)")
