#!/usr/bin/env python3
"""Cycle split of the Winograd-domain weight-gradient kernel (conv3x3_wgrad.hip built with -DWW_STAMP: tools/variant.sh conv3x3_wgrad ww_STAMP -DWW_STAMP):
ND_LIB=tools/_build/lib_ww_STAMP.so [ND_WGRAD_WINO8=0] python tools/wgrad_clock.py     per wave role: cycles per tile group = the MFMA slots / the waits
at the barriers (four-wave form: one barrier; eight-wave form: segments 1, 2 and the waits at B, A); the shader clock under load."""
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
    ws = torch.zeros(n_ws + 4096 * 8 * 8 + 1024, device=DEV)
    for _ in range(3):
        L.call("nd_conv3x3_wgrad_nhwc_f32", x.data_ptr(), cin, gy.data_ptr(), cout, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), B, H, W, cin, cout, st)
    torch.cuda.synchronize()
    groups = B * (H // 4) * (W // 16)
    wide = os.environ.get("ND_WGRAD_WINO8", "1") != "0" and cout % 64 == 0 and groups * (cin // 32) * (cout // 32) >= 20000      # ww_plan: the eight-wave form
    blocks = (cin // 32) * (cout // (64 if wide else 32))
    S = min((-(-blocks * c // 256) * (-(-groups // c) + 14), c) for c in range(1, min(groups, 1024) + 1) if -(-blocks * c // 256) <= 9)[1]      # ww_plan
    base = S * 36 * cout * cin + S * cout
    nw = 8 if wide else 4
    d = ws[base: base + blocks * S * nw * 8].cpu().view(-1, nw, 8).double()
    n = d[:, :, 2]
    mhz = float((d[:, :, 0] / (d[:, :, 1] / 100.0)).mean())
    print((B, H, W, cin, cout), f"{'eight' if wide else 'four'}-wave form, S={S} workgroups={blocks * S} tile groups per workgroup={float(n.mean()):.1f} clock {mhz:.0f} MHz "
          f"kernel {float(d[:, :, 0].mean()):.0f} cycles")
    if wide:
        roles = ("V rows 0-2", "V rows 3-5", "D half 0 rows 0-2", "D half 0 rows 3-5", "halo staging", "halo staging", "D half 1 rows 0-2", "D half 1 rows 3-5")
        for w in range(8):
            print(f"   wave {w} {roles[w]:18s} per group: " + "  ".join(f"{nm} {float((d[:, w, i] / n[:, w]).mean()):5.0f}" for i, nm in
                  ((3, "segment 1"), (4, "wait at B"), (5, "segment 2"), (6, "wait at A"))) + f"   total {float((d[:, w, 3:7].sum(1) / n[:, w]).mean()):5.0f}   (MFMA pipe per SIMD 2304)")
    else:
        for w, name in enumerate(("V rows 0-2", "V rows 3-5", "D rows 0-2", "D rows 3-5")):
            print(f"   wave {w} {name:12s} per group: slots {float((d[:, w, 4] / n[:, w]).mean()):6.0f}  barrier {float((d[:, w, 5] / n[:, w]).mean()):6.0f}  "
                  f"total {float((d[:, w, 4:6].sum(1) / n[:, w]).mean()):6.0f}   (MFMA pipe 1152)")
