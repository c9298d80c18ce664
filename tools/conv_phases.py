"""Diagnostic: phase timestamps of conv3x3 workgroups (build with -DND_STAMP into a side library)."""
import os, sys, subprocess, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
side = os.path.join(REPO, "tools", "_build", "libnd_stamp.so")   # built in the container: see DESIGN.md / tools/README
import torch
torch.zeros(1, device="cuda")
from noisediff_amd import _lib as L
L.LIB_PATH = side
L.load(side)
import numpy as np
import hiputil as hu
ctx = hu.Ctx()
B, H, W, cin, cout = 16, 256, 256, int(os.environ.get("CIN", 64)), 64
x = torch.randn(B, H, W, cin, device=hu.DEV); w = torch.randn(cout, cin, 3, 3) * 0.05
wp = hu.pack_conv3(ctx, w); b = torch.randn(cout, device=hu.DEV); out = torch.empty(B, H, W, cout, device=hu.DEV)
torch.cuda.synchronize()
d = L.Conv3x3(); d.src, d.weight, d.bias, d.out = hu.src(x), wp.data_ptr(), b.data_ptr(), out.data_ptr()
d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
for _ in range(3):
    L.call("nd_conv3x3_nhwc_f32", C.byref(d), ctx.stream)
ctx.sync()
n = 8192 * 8
buf = (C.c_ulonglong * n)()
ctx.lib.nd_dbg_read_stamps.argtypes = [C.c_void_p, C.c_int]
assert ctx.lib.nd_dbg_read_stamps(buf, n) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
t0 = s[:, 0].min()
ph = (s[:, :6] - t0) * 0.01   # us
dur = np.diff(ph, axis=1)
names = ["launch->commit0", "compute0", "commit1", "compute1", "epilogue"]
print("kernel span us:", ph[:, 5].max())
for i, nme in enumerate(names):
    print(f"{nme:16s} mean {dur[:, i].mean():7.2f} us  p10 {np.percentile(dur[:, i], 10):7.2f}  p90 {np.percentile(dur[:, i], 90):7.2f}")
print("WG life mean", (ph[:, 5] - ph[:, 0]).mean())
# co-residency: group by (xcc, hw_id CU bits) and show the first few WG timelines on one CU
hw = s[:, 6]; xcc = s[:, 7] & 0xF
cu_key = (xcc << 16) | ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 0x1) << 8)   # cu_id, se_id, sh_id
k0 = cu_key[0]
idx = np.where(cu_key == k0)[0]
idx = idx[np.argsort(ph[idx, 0])][:12]
for i in idx:
    print("WG", i, "simd/wave", (hw[i] >> 4) & 3, hw[i] & 0xF, " ".join(f"{v:8.2f}" for v in ph[i]))
