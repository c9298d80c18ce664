"""Diagnostic: phase split of pointwise_big_kernel / pointwise_split_kernel (needs a -DPWB_STAMP -DPWS_STAMP side build of pointwise.hip, ND_LIB).  FORM=fp32|split; RES=1 adds a residual."""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
torch.zeros(1, device="cuda")
from noisediff_amd import _lib as L
L.load(os.environ["ND_LIB"])
import hiputil as hu
ctx = hu.Ctx()
B = 16
split = os.environ.get("FORM", "fp32") == "split"
entry = "nd_pointwise_gemm_split_nhwc_f32" if split else "nd_pointwise_gemm_nhwc_f32"
for (HW, cin, cout) in [(1024, 1024, 512), (1024, 512, 512), (4096, 256, 256), (16384, 128, 128), (16384, 256, 128), (16384, 128, 256)]:
    x = torch.randn(B, HW, cin, device=hu.DEV); w = torch.randn(cout, cin) / cin ** 0.5
    if split:
        wd = hu.dev(w); wp = torch.empty(ctx.lib.nd_pack_pointwise_weight_split_floats(cin, cout), device=hu.DEV)
        L.call("nd_pack_pointwise_weight_split", wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream); ctx.sync()
    else:
        wp = hu.pack_pw(ctx, w)
    out = torch.zeros(B, HW, cout, device=hu.DEV)
    d = L.Pointwise(); d.src, d.weight, d.out = hu.src(x), wp.data_ptr(), out.data_ptr()
    d.B, d.HW, d.W, d.cin, d.cout, d.ldo = B, HW, int(HW ** 0.5), cin, cout, cout
    if os.environ.get("RES"):
        res = torch.randn(B, HW, cout, device=hu.DEV); d.res0, d.ldr0 = res.data_ptr(), cout
    torch.cuda.synchronize()
    for _ in range(3):
        L.call(entry, C.byref(d), ctx.stream); ctx.sync()
    o = out.view(B, HW // 128, 128, cout // 128, 128)[:, :, 0, :, :8].contiguous().view(torch.int64).view(-1, 4).double().cpu()
    n_chunks = cin // (32 if split else 64)
    oi = out.view(B, HW // 128, 128, cout // 128, 128)[:, :, 0, :, :8].contiguous().view(torch.int64).view(-1, 4)[:, 3].cpu()
    ph = [float(((oi >> (16 * i)) & 0xFFFF).double().mean()) for i in range(4)]
    print((HW, cin, cout), f"tiles {o.shape[0]}: prologue {o[:, 0].mean():.0f} cycles, K loop {o[:, 1].mean():.0f} = {o[:, 1].mean() / n_chunks:.0f} per chunk (MFMA {1536 if split else 8192}),",
          f"epilogue {o[:, 2].mean():.0f} = barrier {ph[0]:.0f} + acc->LDS {ph[1]:.0f} + barrier {ph[2]:.0f} + rows->global {ph[3]:.0f}", flush=True)
