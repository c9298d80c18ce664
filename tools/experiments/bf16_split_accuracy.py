#!/usr/bin/env python3
"""CPU emulation: how accurate is the F(4x4,3x3) convolution when its position products M = sum_c U V run on bf16 matrix instructions with fp32 accumulation and
the operands split into bf16 terms (V = V1 + V2 + V3, U = U1 + U2 + U3; 3 terms kept: V1U1 + V1U2 + V2U1; 6 terms: + V2U2 + V1U3 + V3U1)?
Against an fp64 direct convolution, next to the fp32 F(4x4) path this build runs today.  Evidence for DESIGN section 8 ("what comes next"): the fp32 MFMA shares the
VALU lanes and the conv kernels sit on an issue-bound plateau; bf16 MFMAs run 16x faster on their own pipe.  No GPU needed."""
import numpy as np, torch
torch.manual_seed(0)
BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)


def bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def split3(t):
    a = bf16(t); r = t - a
    b = bf16(r); r = r - b
    return a, b, bf16(r)


def run(cin, cout, H, act_scale=1.0):
    x = (torch.randn(1, cin, H, H) * act_scale).float()
    bound = 1.0 / np.sqrt(9 * cin)
    w = ((torch.rand(cout, cin, 3, 3) * 2 - 1) * bound).float()
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    T = H // 4
    d = xp.unfold(2, 6, 4).unfold(3, 6, 4)                                  # [1, cin, T, T, 6, 6]
    V = torch.einsum("ia,bctuaj,kj->bctuik", BT.float(), d, BT.float())     # fp32 input transform
    U = torch.einsum("ia,ocaj,kj->ocik", G, w.double(), G).float()          # exact-rounded once, as the pack kernel does

    def finish(M):                                                            # M [1, cout, T, T, 6, 6] fp32
        Y = torch.einsum("ia,botuaj,kj->botuik", AT.float(), M, AT.float())
        return Y.permute(0, 1, 2, 4, 3, 5).reshape(1, cout, H, H)

    prod = lambda Vx, Ux: torch.einsum("bctuik,ocik->botuik", Vx, Ux)        # fp32 accumulation over cin
    res = {"fp32 (today)": finish(prod(V, U))}
    V1, V2, V3 = split3(V); U1, U2, U3 = split3(U)
    res["bf16 x1 (plain bf16)"] = finish(prod(V1, U1))
    m3 = prod(V1, U1) + prod(V1, U2) + prod(V2, U1)
    res["bf16 split, 3 products"] = finish(m3)
    res["bf16 split, 6 products"] = finish(m3 + prod(V2, U2) + prod(V1, U3) + prod(V3, U1))
    scale = float(ref.abs().max())
    return {k: float((v.double() - ref).abs().max()) / max(1.0, scale) for k, v in res.items()}, scale


for cin, cout, H in ((64, 64, 32), (256, 64, 16), (512, 64, 16)):
    errs, scale = run(cin, cout, H, act_scale=1.0)
    print(f"cin {cin:4d} cout {cout} {H}x{H}, |y|max {scale:.2f}: " + "  ".join(f"{k}: {v:.2e}" for k, v in errs.items()))
