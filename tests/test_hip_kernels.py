"""GPU parity of every C-ABI kernel against plain torch CPU ops (the oracle's building blocks).

Each case calls libnoisediff_hip.so directly through ctypes; torch is only the allocator.
Tolerance: 1e-3 relative fp32 is the north-star bound; kernels are exact-fp32 so tests use 2e-5
(max-abs over max(1, |ref|)) unless a comment says otherwise.
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from noisediff_amd import _lib as L, synth
from oracle import noisediff_oracle as O
from util import rel_err

TOL = 2e-5


@pytest.fixture(scope="module")
def ctx():
    import hiputil
    return hiputil.Ctx()


def U(name, shape, lo=-1.0, hi=1.0):
    return synth.uniform(11, name, shape, lo, hi)


def test_library_is_gfx950(ctx):
    buf = C.create_string_buffer(64)
    L.call("nd_device_arch", buf, 64)
    assert buf.value.decode().startswith("gfx950")


# (B, H, W, cin, cout) -- chosen so that every tiling of conv3x3.hip's choose_tiling is exercised:
CONV_CASES = {
    "t8x8_bn64_masks": (2, 24, 40, 16, 16),        # (8,1,1): ragged edges in y and x, cout < 64
    "t8x8_bn64_partial_chunk": (2, 16, 16, 48, 48),  # cin = 32 + 16 (partial K chunk), cout mask
    "t16_bn128": (8, 64, 64, 32, 256),             # (16,2,2)
    "t16_bn64": (16, 56, 72, 16, 64),              # (16,2,1) with a ragged right edge
    "t8_bn128": (64, 8, 8, 16, 1024),              # (8,1,2)
    "tiny_4x4": (2, 4, 4, 64, 128),                # image smaller than a tile
}


@pytest.mark.parametrize("case", sorted(CONV_CASES))
def test_conv3x3_plain_and_stats(ctx, case):
    import hiputil as hu
    B, H, W, cin, cout = CONV_CASES[case]
    x = U(case + ".x", (B, cin, H, W), -1.5, 1.5)
    w = U(case + ".w", (cout, cin, 3, 3), -0.2, 0.2)
    b = U(case + ".b", (cout,))
    ref = F.conv2d(x, w, b, padding=1)
    out, st, sc, slots = hu.conv3x3(ctx, hu.src(hu.nhwc(x)), hu.pack_conv3(ctx, w), hu.dev(b), B, H, W, cin, cout, stats=True)
    got = hu.nchw(out)
    assert not torch.isnan(got).any()
    assert rel_err(got, ref) < TOL
    # GroupNorm statistics through the finalize kernel == torch's group_norm of the same tensor
    groups = 8 if cout % 8 == 0 else 2
    gamma, beta = U(case + ".g", (cout,), 0.5, 1.5), U(case + ".be", (cout,))
    mad = hu.gn_finalize(ctx, st, sc, slots, hu.dev(gamma), hu.dev(beta), None, B, cout, groups).cpu()
    gn = F.group_norm(ref, groups, gamma, beta, eps=1e-5)
    mine = (ref - mad[:, 0, :, None, None]) * mad[:, 1, :, None, None] + mad[:, 2, :, None, None]
    assert rel_err(mine, gn) < TOL


def test_conv3x3_concat_upsample_and_prologues(ctx):
    import hiputil as hu
    B, H, W = 2, 16, 24
    # virtual concat == torch.cat(dim=1)
    xa, xb = U("cc.a", (B, 16, H, W)), U("cc.b", (B, 32, H, W))
    w, b = U("cc.w", (24, 48, 3, 3), -0.2, 0.2), U("cc.bias", (24,))
    ref = F.conv2d(torch.cat((xa, xb), 1), w, b, padding=1)
    out, *_ = hu.conv3x3(ctx, hu.src(hu.nhwc(xa), hu.nhwc(xb)), hu.pack_conv3(ctx, w), hu.dev(b), B, H, W, 48, 24)
    assert rel_err(hu.nchw(out), ref) < TOL
    # nearest x2 upsample folded into addressing == nn.Upsample + conv
    xs = U("up.x", (B, 16, H // 2, W // 2))
    w, b = U("up.w", (8, 16, 3, 3), -0.2, 0.2), U("up.b", (8,))
    ref = F.conv2d(F.interpolate(xs, scale_factor=2, mode="nearest"), w, b, padding=1)
    out, *_ = hu.conv3x3(ctx, hu.src(hu.nhwc(xs), upsample=1), hu.pack_conv3(ctx, w), hu.dev(b), B, H, W, 16, 8)
    assert rel_err(hu.nchw(out), ref) < TOL
    # affine + SiLU prologue: conv(silu((x - M) * A + D)), zero padding applied AFTER the activation
    x = U("af.x", (B, 16, H, W), -2, 2)
    M, A, D = U("af.M", (B, 16)), U("af.A", (B, 16), 0.5, 1.5), U("af.D", (B, 16))
    w, b = U("af.w", (16, 16, 3, 3), -0.2, 0.2), U("af.b", (16,))
    act = F.silu((x - M[:, :, None, None]) * A[:, :, None, None] + D[:, :, None, None])
    ref = F.conv2d(act, w, b, padding=1)
    mad = hu.dev(torch.stack((M, A, D), 1))
    out, *_ = hu.conv3x3(ctx, hu.src(hu.nhwc(x), None, L.PRO_AFFINE_SILU, mad=mad), hu.pack_conv3(ctx, w), hu.dev(b), B, H, W, 16, 16)
    assert rel_err(hu.nchw(out), ref) < TOL
    # ... plus per-pixel scale/shift maps (ResnetBlock2)
    sc, sh = U("af.sc", (B, 16, H, W), -0.5, 0.5), U("af.sh", (B, 16, H, W), -0.5, 0.5)
    act = F.silu(((x - M[:, :, None, None]) * A[:, :, None, None] + D[:, :, None, None]) * (sc + 1) + sh)
    ref = F.conv2d(act, w, b, padding=1)
    mp = hu.nhwc(torch.cat((sc, sh), 1))
    out, *_ = hu.conv3x3(ctx, hu.src(hu.nhwc(x), None, L.PRO_AFFINE_MAP_SILU, mad=mad, map=mp), hu.pack_conv3(ctx, w), hu.dev(b), B, H, W, 16, 16)
    assert rel_err(hu.nchw(out), ref) < TOL


@pytest.mark.parametrize("entry", ["nd_conv3x3_wino_nhwc_f32", "nd_conv3x3_wino2_nhwc_f32"])
def test_conv3x3_winograd_addressing_and_prologues(ctx, entry):
    """Both Winograd kernels: nearest-x2 upsample addressing, the per-pixel scale/shift map (ResnetBlock2), the LSID
    LeakyReLU prologues (whole input / second concat source only), on image sizes that are not multiples of the tile."""
    import hiputil as hu

    def run(s, w, b, B, H, W):
        cout, cin = w.shape[:2]
        wd = hu.dev(w)
        wp = hu.full((ctx.lib.nd_pack_conv3x3_wino_weight_floats(cin, cout),))
        L.call("nd_pack_conv3x3_wino_weight", wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream)
        out, bd = hu.full((B, H, W, cout)), hu.dev(b)
        d = L.Conv3x3()
        d.src, d.weight, d.bias, d.out = s, wp.data_ptr(), bd.data_ptr(), out.data_ptr()
        d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
        L.call(entry, C.byref(d), ctx.stream)
        ctx.sync()
        return hu.nchw(out)

    B, H, W = 2, 36, 52                                    # 3 x 4 tiles, ragged on both axes
    xs = U("wa.up", (B, 32, H // 2, W // 2))
    w, b = U("wa.w", (40, 32, 3, 3), -0.2, 0.2), U("wa.b", (40,))
    ref = F.conv2d(F.interpolate(xs, scale_factor=2, mode="nearest"), w, b, padding=1)
    assert rel_err(run(hu.src(hu.nhwc(xs), upsample=1), w, b, B, H, W), ref) < 1e-5
    x = U("wa.x", (B, 32, H, W), -2, 2)
    M, A, D = U("wa.M", (B, 32)), U("wa.A", (B, 32), 0.5, 1.5), U("wa.D", (B, 32))
    sc, sh = U("wa.sc", (B, 32, H, W), -0.5, 0.5), U("wa.sh", (B, 32, H, W), -0.5, 0.5)
    act = F.silu(((x - M[:, :, None, None]) * A[:, :, None, None] + D[:, :, None, None]) * (sc + 1) + sh)
    s = hu.src(hu.nhwc(x), None, L.PRO_AFFINE_MAP_SILU, mad=hu.dev(torch.stack((M, A, D), 1)), map=hu.nhwc(torch.cat((sc, sh), 1)))
    assert rel_err(run(s, w, b, B, H, W), F.conv2d(act, w, b, padding=1)) < 1e-5
    assert rel_err(run(hu.src(hu.nhwc(x), None, L.PRO_LEAKY), w, b, B, H, W), F.conv2d(F.leaky_relu(x, 0.2), w, b, padding=1)) < 1e-5
    x2 = U("wa.x2", (B, 32, H, W), -2, 2)                  # cat(up, skip): LeakyReLU on the skip only (SID_arch.py:135-139)
    w2 = U("wa.w2", (24, 64, 3, 3), -0.2, 0.2)
    ref = F.conv2d(torch.cat((x, F.leaky_relu(x2, 0.2)), 1), w2, b[:24], padding=1)
    assert rel_err(run(hu.src(hu.nhwc(x), hu.nhwc(x2), L.PRO_LEAKY_SECOND), w2, b[:24], B, H, W), ref) < 1e-5


def test_conv3x3_wino2_is_bitwise_repeatable(ctx):
    """The kernel issues its MFMAs through inline asm, outside the compiler's hazard bookkeeping: a missed wait state would
    show as run-to-run differences.  Same launch five times -> identical bits (outputs and GroupNorm partials)."""
    import hiputil as hu
    B, H, W, cin, cout = 4, 64, 64, 64, 64
    x, w, b = U("rp.x", (B, cin, H, W), -1.5, 1.5), U("rp.w", (cout, cin, 3, 3), -0.2, 0.2), U("rp.b", (cout,))
    M, A, D = U("rp.M", (B, cin)), U("rp.A", (B, cin), 0.5, 1.5), U("rp.D", (B, cin))
    xd, wd, bd, mad = hu.nhwc(x), hu.dev(w), hu.dev(b), hu.dev(torch.stack((M, A, D), 1))
    wp = hu.full((ctx.lib.nd_pack_conv3x3_wino_weight_floats(cin, cout),))
    L.call("nd_pack_conv3x3_wino_weight", wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream)
    slots = ctx.lib.nd_conv3x3_wino_stat_slots(H, W)
    runs = []
    for _ in range(5):
        out, st, sc = hu.full((B, H, W, cout)), hu.full((B, slots, cout, 2)), hu.full((slots,))
        d = L.Conv3x3()
        d.src = hu.src(xd, None, L.PRO_AFFINE_SILU, mad=mad)
        d.weight, d.bias, d.out, d.stats, d.slot_count = wp.data_ptr(), bd.data_ptr(), out.data_ptr(), st.data_ptr(), sc.data_ptr()
        d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
        L.call("nd_conv3x3_wino2_nhwc_f32", C.byref(d), ctx.stream)
        ctx.sync()
        runs.append((out.cpu(), st.cpu()))
    assert torch.isfinite(runs[0][0]).all() and torch.isfinite(runs[0][1]).all()
    assert all(torch.equal(runs[0][0], r[0]) and torch.equal(runs[0][1], r[1]) for r in runs[1:])


def test_conv3x3_rejects_bad_arguments(ctx):
    import hiputil as hu
    x = hu.nhwc(U("bad.x", (1, 12, 8, 8)))
    d = L.Conv3x3()
    d.src = hu.src(x)
    d.weight = d.out = x.data_ptr()
    d.B, d.H, d.W, d.cin, d.cout, d.ldo = 1, 8, 8, 12, 8, 8
    assert ctx.lib.nd_conv3x3_nhwc_f32(C.byref(d), ctx.stream) == -2     # ND_E_SHAPE: cin % 8
    assert b"multiple of 8" in ctx.lib.nd_last_error()
    assert ctx.lib.nd_conv3x3_nhwc_f32(None, ctx.stream) == -1            # ND_E_BADARG


PW_CASES = {  # (B, HW, W, cin, cout)
    "c64": (2, 32 * 32, 32, 64, 64),
    "k8_n64": (2, 16 * 16, 16, 8, 64),
    "k24_n16": (1, 16 * 16, 16, 24, 16),
    "n4": (2, 24 * 40, 40, 64, 4),
    "wide": (16, 64 * 32, 32, 128, 256),          # (2,2) tiling
    "c64_big": (64, 32 * 32, 32, 64, 64),         # (2,1) tiling
    "ragged": (3, 25, 5, 192, 96),                # HW < tile, 3 K chunks
}


@pytest.mark.parametrize("case", sorted(PW_CASES))
def test_pointwise_plain_and_epilogues(ctx, case):
    import hiputil as hu
    B, HW, W, cin, cout = PW_CASES[case]
    x = U(case + ".x", (B, HW, cin), -1.5, 1.5)
    w, b = U(case + ".w", (cout, cin), -0.3, 0.3), U(case + ".b", (cout,))
    r0, r1, vec = U(case + ".r0", (B, HW, cout)), U(case + ".r1", (B, HW, cout)), U(case + ".v", (B, cout))
    wp = hu.pack_pw(ctx, w)
    out = hu.pointwise(ctx, hu.src(hu.dev(x)), wp, hu.dev(b), B, HW, W, cin, cout)
    assert rel_err(out.cpu(), F.linear(x, w, b)) < TOL
    out = hu.pointwise(ctx, hu.src(hu.dev(x)), wp, hu.dev(b), B, HW, W, cin, cout, act=L.ACT_GELU,
                       res0=hu.dev(r0), res1=hu.dev(r1), vec=hu.dev(vec))
    assert rel_err(out.cpu(), F.gelu(F.linear(x, w, b)) + r0 + r1 + vec[:, None, :]) < TOL
    # fused ResnetBlock tail: W x + b + silu((t - M) * A + D)
    t = U(case + ".t", (B, HW, cout), -2, 2)
    mad = U(case + ".mad", (B, 3, cout), 0.5, 1.5)
    ref = F.linear(x, w, b) + F.silu((t - mad[:, None, 0]) * mad[:, None, 1] + mad[:, None, 2])
    out = hu.pointwise(ctx, hu.src(hu.dev(x)), wp, hu.dev(b), B, HW, W, cin, cout, gn_t=hu.dev(t), gn_mad=hu.dev(mad))
    assert rel_err(out.cpu(), ref) < TOL


@pytest.mark.parametrize("cin,cout,c0", [(64, 4, 0), (64, 8, 0), (128, 4, 64), (24, 4, 0), (256, 8, 128)])
def test_pointwise_narrow_outputs_streaming_kernel(ctx, cin, cout, c0):
    """cout <= 8 (final_conv, Diffusion_arch.py:554: 64 -> 4): nd_pointwise_gemm_nhwc_f32 runs a streaming dot product (four lanes per pixel) instead of an MFMA tile
    that is 15/16 padding.  Plain + bias, a virtual concat, activation + two residuals, no bias, a pixel count that is not a multiple of 64, bitwise repeat; with a
    per-sample vector the layer goes back to the MFMA kernels (same result)."""
    import hiputil as hu
    B, HW, W = 3, 37 * 5, 5                                   # 555 pixels: the last block of 64 is ragged
    x = U(f"nar.x.{cin}", (B, HW, cin), -1.5, 1.5)
    w, b = U(f"nar.w.{cin}.{cout}", (cout, cin), -0.3, 0.3), U(f"nar.b.{cout}", (cout,))
    r0, r1, vec = U(f"nar.r0.{cout}", (B, HW, cout)), U(f"nar.r1.{cout}", (B, HW, cout)), U(f"nar.v.{cout}", (B, cout))
    wp, bd = hu.pack_pw(ctx, w), hu.dev(b)
    s = hu.src(hu.dev(x[..., :c0].contiguous()), hu.dev(x[..., c0:].contiguous())) if c0 else hu.src(hu.dev(x))
    lin = F.linear(x, w, b)
    out = hu.pointwise(ctx, s, wp, bd, B, HW, W, cin, cout)
    assert rel_err(out.cpu(), lin) < TOL
    out2 = hu.pointwise(ctx, s, wp, bd, B, HW, W, cin, cout)
    assert torch.equal(out.cpu(), out2.cpu())
    out = hu.pointwise(ctx, s, wp, None, B, HW, W, cin, cout, act=L.ACT_SILU, res0=hu.dev(r0), res1=hu.dev(r1))
    assert rel_err(out.cpu(), F.silu(F.linear(x, w)) + r0 + r1) < TOL
    out = hu.pointwise(ctx, s, wp, bd, B, HW, W, cin, cout, act=L.ACT_GELU, res0=hu.dev(r0))
    assert rel_err(out.cpu(), F.gelu(lin) + r0) < TOL
    out = hu.pointwise(ctx, s, wp, bd, B, HW, W, cin, cout, vec=hu.dev(vec))          # (a per-sample vector: the MFMA kernels)
    assert rel_err(out.cpu(), lin + vec[:, None]) < TOL


CHAIN_CASES = {  # (B, HW, [widths]): Mlp chains (two stages) and AttnBlock tails (three stages, LayerNorm + residuals)
    "mlp1_d64": (2, 64, [8, 64, 64]),
    "mlp2_d64": (2, 96, [64, 64, 64]),
    "mlp3_d64": (3, 32, [64, 64, 4]),
    "mlp_d16": (2, 64, [16, 16, 16]),
    "mlp_d48": (2, 64, [48, 48, 48]),
    "ff_d64": (3, 160, [64, 128, 64, 64]),
    "ff_d48": (2, 64, [48, 96, 48, 48]),
    "ff_d32": (2, 64, [32, 64, 32, 32]),
    "ff_d16": (2, 32, [16, 32, 16, 16]),
}


@pytest.mark.parametrize("form", ["fp32", "split"])
@pytest.mark.parametrize("case", sorted(CHAIN_CASES))
def test_pointwise_chain_matches_layer_by_layer(ctx, case, form):
    """nd_pointwise_chain_nhwc_f32 / nd_pointwise_chain_split_nhwc_f32 (bf16 x 3 split products, r6) == the same Linear layers applied one by one
    (Mlp :340-356, AttnBlock tail :405-443), at the same tolerance."""
    import hiputil as hu
    B, HW, widths = CHAIN_CASES[case]
    pack, entry = (("nd_pack_chain_weight", "nd_pointwise_chain_nhwc_f32") if form == "fp32" else
                   ("nd_pack_chain_weight_split", "nd_pointwise_chain_split_nhwc_f32"))
    n = len(widths) - 1
    x = U(case + ".x", (B, HW, widths[0]), -1.5, 1.5)
    ws = [U(f"{case}.w{i}", (widths[i + 1], widths[i]), -0.3, 0.3) for i in range(n)]
    bs = [U(f"{case}.b{i}", (widths[i + 1],)) for i in range(n)]
    keep = []
    d = L.Chain()
    for i in range(n):
        wd = hu.dev(ws[i])
        wp = hu.full((getattr(ctx.lib, pack + "_floats")(widths[i], widths[i + 1], int(i == 0)),))
        L.call(pack, wd.data_ptr(), wp.data_ptr(), widths[i], widths[i + 1], int(i == 0), ctx.stream)
        bd = hu.dev(bs[i])
        keep += [wd, wp, bd]
        d.st[i].weight, d.st[i].bias, d.st[i].cin, d.st[i].cout = wp.data_ptr(), bd.data_ptr(), widths[i], widths[i + 1]
    ctx.sync()
    out = hu.full((B, HW, widths[-1]))
    d.out, d.n_stages, d.B, d.HW, d.ldo = out.data_ptr(), n, B, HW, widths[-1]
    if n == 2:          # Mlp: fc2(GELU(fc1(x))); the 8-wide case is the virtual concat cat[clean_img, x] of shot_mlp1
        s = hu.src(hu.dev(x[..., :4].contiguous()), hu.dev(x[..., 4:].contiguous())) if widths[0] == 8 else hu.src(hu.dev(x))
        d.src = s
        d.st[0].act = L.ACT_GELU
        ref = F.linear(F.gelu(F.linear(x, ws[0], bs[0])), ws[1], bs[1])
    else:               # AttnBlock tail
        Cc = widths[0]
        vec, g, be = U(case + ".v", (B, Cc)), U(case + ".g", (Cc,), 0.5, 1.5), U(case + ".be", (Cc,))
        s = hu.src(hu.dev(x), None, L.PRO_LAYERNORM, vec=hu.dev(vec), gamma=hu.dev(g), beta=hu.dev(be))
        d.src = s
        d.st[0].act, d.st[1].res, d.st[2].res = L.ACT_GELU, L.CHAIN_RES_INPUT, L.CHAIN_RES_INPUT_RAW
        x1 = x + vec[:, None]
        h = F.gelu(F.linear(F.layer_norm(x1, (Cc,), g, be, eps=1e-5), ws[0], bs[0]))
        ref = F.linear(F.linear(h, ws[1], bs[1]) + x1, ws[2], bs[2]) + x
    L.call(entry, C.byref(d), ctx.stream)
    ctx.sync()
    assert rel_err(out.cpu(), ref) < TOL
    keep.append(s)                                      # (the descriptor holds a COPY of the struct: the tensors behind its pointers live in s._refs)
    first = out.clone()
    torch.cuda.synchronize()
    L.call(entry, C.byref(d), ctx.stream)
    ctx.sync()
    assert torch.equal(out, first)                      # bitwise repeatable
    # widths this build does not instantiate are refused, not approximated
    d.st[0].cout = d.st[1].cin = 160
    assert getattr(ctx.lib, entry)(C.byref(d), ctx.stream) != 0
    assert b"not instantiated" in ctx.lib.nd_last_error()


SPLIT_GEMM_CASES = {  # (B, HW, cin, cout, first concat source, kind): the wide 1x1 layers of the U-Net (Diffusion_arch.py:156,405-443,252-253)
    "ff1_ln_gelu": (2, 200, 128, 256, 0, "ln"),      # LayerNorm(x + v) -> Linear -> GELU, ragged last pixel tile
    "ff2_res_vec": (2, 256, 256, 128, 0, "res"),     # Linear + x + v
    "res_conv_cat": (3, 128, 192, 128, 128, "gn"),   # res_conv over the virtual concat + silu(GN(c2)) tail
    "qkv_nobias": (1, 1024, 512, 384, 0, "plain"),
    "two_chunks": (2, 64, 64, 128, 0, "plain"),
}


@pytest.mark.parametrize("case", sorted(SPLIT_GEMM_CASES))
def test_pointwise_split_matches_torch_and_the_fp32_kernel(ctx, case):
    """nd_pointwise_gemm_split_nhwc_f32 (r6: three bf16 terms per fp32 operand, six products on v_mfma_f32_32x32x16_bf16, fp32 accumulation): the same
    prologues / epilogues as nd_pointwise_gemm_nhwc_f32, the same tolerance against torch, error against fp64 no worse than 1.5x the fp32 kernel's."""
    import hiputil as hu
    B, HW, cin, cout, c0, kind = SPLIT_GEMM_CASES[case]
    x = U(case + ".x", (B, HW, cin), -1.5, 1.5)
    w, b = U(case + ".w", (cout, cin), -0.2, 0.2), U(case + ".b", (cout,))
    wp = hu.pack_pw(ctx, w)
    wd, ws = hu.dev(w), hu.full((ctx.lib.nd_pack_pointwise_weight_split_floats(cin, cout),))
    L.call("nd_pack_pointwise_weight_split", wd.data_ptr(), ws.data_ptr(), cin, cout, ctx.stream)
    ctx.sync()
    kw, bias = {}, hu.dev(b)
    x64, w64, b64 = x.double(), w.double(), b.double()
    if kind == "ln":
        vec, g, be = U(case + ".v", (B, cin)), U(case + ".g", (cin,), 0.5, 1.5), U(case + ".be", (cin,))
        xd, vd = hu.dev(x), hu.dev(vec)
        rs = hu.full((B, HW, 2))
        L.call("nd_layernorm_stats_f32", xd.data_ptr(), cin, vd.data_ptr(), rs.data_ptr(), B, HW, cin, 1e-5, ctx.stream)
        ctx.sync()
        s = hu.src(xd, None, L.PRO_LAYERNORM, vec=vd, gamma=hu.dev(g), beta=hu.dev(be), rowstats=rs)
        kw["act"] = L.ACT_GELU
        ref = F.gelu(F.linear(F.layer_norm(x64 + vec.double()[:, None], (cin,), g.double(), be.double(), eps=1e-5), w64, b64))
    elif kind == "res":
        r0, vec = U(case + ".r0", (B, HW, cout)), U(case + ".v", (B, cout))
        s = hu.src(hu.dev(x))
        kw.update(res0=hu.dev(r0), vec=hu.dev(vec))
        ref = F.linear(x64, w64, b64) + r0.double() + vec.double()[:, None]
    elif kind == "gn":
        t, mad = U(case + ".t", (B, HW, cout), -2, 2), U(case + ".mad", (B, 3, cout), 0.5, 1.5)
        s = hu.src(hu.dev(x[..., :c0].contiguous()), hu.dev(x[..., c0:].contiguous()))
        kw.update(gn_t=hu.dev(t), gn_mad=hu.dev(mad))
        m64 = mad.double()
        ref = F.linear(x64, w64, b64) + F.silu((t.double() - m64[:, None, 0]) * m64[:, None, 1] + m64[:, None, 2])
    else:
        s = hu.src(hu.dev(x))
        if case == "qkv_nobias":
            bias, b64 = None, None
        ref = F.linear(x64, w64, b64)
    d = L.Pointwise()
    d.src, d.cin, d.cout = s, cin, cout
    assert ctx.lib.nd_pointwise_gemm_split_takes(C.byref(d)) == 1
    o32 = hu.pointwise(ctx, s, wp, bias, B, HW, int(HW ** 0.5), cin, cout, **kw)
    osp = hu.pointwise(ctx, s, ws, bias, B, HW, int(HW ** 0.5), cin, cout, entry="nd_pointwise_gemm_split_nhwc_f32", **kw)
    assert rel_err(osp.cpu(), ref.float()) < TOL
    e32, esp = (o32.cpu().double() - ref).abs().max().item(), (osp.cpu().double() - ref).abs().max().item()
    assert esp <= 1.5 * e32 + 1e-7, (esp, e32)
    again = hu.pointwise(ctx, s, ws, bias, B, HW, int(HW ** 0.5), cin, cout, entry="nd_pointwise_gemm_split_nhwc_f32", **kw)
    assert torch.equal(osp, again)                      # bitwise repeatable
    # shapes outside the kernel's set are refused by the split entry, not approximated
    d2 = L.Pointwise()
    d2.src, d2.cin, d2.cout = s, cin, 64
    assert ctx.lib.nd_pointwise_gemm_split_takes(C.byref(d2)) == 0


def test_pointwise_prologues(ctx):
    import hiputil as hu
    B, H, W, Cc = 2, 16, 16, 64
    x = U("pp.x", (B, H * W, Cc), -2, 2)
    vec = U("pp.vec", (B, Cc))
    g, be = U("pp.g", (Cc,), 0.5, 1.5), U("pp.be", (Cc,))
    w, b = U("pp.w", (128, Cc), -0.3, 0.3), U("pp.b", (128,))
    wp = hu.pack_pw(ctx, w)
    # LayerNorm(x + vec) prologue (AttnBlock.norm2 on x + cross-attention bias)
    ref = F.linear(F.layer_norm(x + vec[:, None], (Cc,), g, be, eps=1e-5), w, b)
    s = hu.src(hu.dev(x), None, L.PRO_LAYERNORM, vec=hu.dev(vec), gamma=hu.dev(g), beta=hu.dev(be))
    assert rel_err(hu.pointwise(ctx, s, wp, hu.dev(b), B, H * W, W, Cc, 128).cpu(), ref) < TOL
    # wide LayerNorm (C = 512 -> two float4 per lane)
    xw, gw, bw = U("pp.xw", (B, 64, 512), -2, 2), U("pp.gw", (512,), 0.5, 1.5), U("pp.bw", (512,))
    ww = U("pp.ww", (64, 512), -0.1, 0.1)
    ref = F.linear(F.layer_norm(xw, (512,), gw, bw, eps=1e-5), ww)
    xwd = hu.dev(xw)
    rs = hu.full((B, 64, 2))
    L.call("nd_layernorm_stats_f32", xwd.data_ptr(), 512, None, rs.data_ptr(), B, 64, 512, 1e-5, ctx.stream)
    ctx.sync()
    mu, var = xw.mean(-1), xw.var(-1, unbiased=False)
    assert rel_err(rs[..., 0].cpu(), mu) < TOL and rel_err(rs[..., 1].cpu(), (var + 1e-5).rsqrt()) < TOL
    s = hu.src(xwd, None, L.PRO_LAYERNORM, gamma=hu.dev(gw), beta=hu.dev(bw), rowstats=rs)
    assert rel_err(hu.pointwise(ctx, s, hu.pack_pw(ctx, ww), None, B, 64, 8, 512, 64).cpu(), ref) < TOL
    # a wide LayerNorm without the statistics is refused, not guessed
    s2 = hu.src(xwd, None, L.PRO_LAYERNORM, gamma=hu.dev(gw), beta=hu.dev(bw))
    d = L.Pointwise()
    d.src, d.weight, d.out = s2, xwd.data_ptr(), xwd.data_ptr()
    d.B, d.HW, d.W, d.cin, d.cout, d.ldo = B, 64, 8, 512, 64, 64
    assert ctx.lib.nd_pointwise_gemm_nhwc_f32(C.byref(d), ctx.stream) == -1
    # narrow LayerNorm with C = 48 (three of the sixteen row lanes idle)
    x48, g48, b48, v48 = U("pp.x48", (B, 100, 48), -2, 2), U("pp.g48", (48,), 0.5, 1.5), U("pp.b48", (48,)), U("pp.v48", (B, 48))
    w48 = U("pp.w48", (96, 48), -0.3, 0.3)
    ref = F.linear(F.layer_norm(x48 + v48[:, None], (48,), g48, b48, eps=1e-5), w48)
    s = hu.src(hu.dev(x48), None, L.PRO_LAYERNORM, vec=hu.dev(v48), gamma=hu.dev(g48), beta=hu.dev(b48))
    assert rel_err(hu.pointwise(ctx, s, hu.pack_pw(ctx, w48), None, B, 100, 10, 48, 96).cpu(), ref) < TOL
    # SiLU prologue
    ref = F.linear(F.silu(x), w, b)
    assert rel_err(hu.pointwise(ctx, hu.src(hu.dev(x), None, L.PRO_SILU), wp, hu.dev(b), B, H * W, W, Cc, 128).cpu(), ref) < TOL
    # virtual concat (shot_mlp1: cat[clean_img, x])
    a4, b4 = U("pp.a4", (B, H * W, 4)), U("pp.b4", (B, H * W, 4))
    w8 = U("pp.w8", (64, 8), -0.3, 0.3)
    ref = F.linear(torch.cat((a4, b4), -1), w8)
    assert rel_err(hu.pointwise(ctx, hu.src(hu.dev(a4), hu.dev(b4)), hu.pack_pw(ctx, w8), None, B, H * W, W, 8, 64).cpu(), ref) < TOL
    # pixel-unshuffle addressing == einops 'b c (h p1) (w p2) -> b (c p1 p2) h w' + conv1x1 (Downsample)
    xc = U("pp.xc", (B, 16, H, W))
    wd, bd = U("pp.wd", (32, 64, 1, 1), -0.3, 0.3), U("pp.bd", (32,))
    sd = {"d.1.weight": wd, "d.1.bias": bd}
    ref = O.pixel_unshuffle_conv(sd, "d", xc)
    s = hu.src(hu.nhwc(xc), None, L.PRO_NONE, unshuffle=1, c0=64, ld0=16)
    out = hu.pointwise(ctx, s, hu.pack_pw(ctx, wd, unshuffle_c=16), hu.dev(bd), B, (H // 2) * (W // 2), W // 2, 64, 32)
    assert rel_err(out.view(B, H // 2, W // 2, 32).permute(0, 3, 1, 2).cpu(), ref) < TOL


@pytest.mark.parametrize("cin", [96, 160, 224, 64])
def test_pointwise_pipelined_chunks_concat_and_prologues(ctx, cin):
    """The software-pipelined kernel (cin % 32 == 0): odd and even numbers of 32-channel chunks (the loop is unrolled by
    two with a peeled tail), a concat boundary inside a chunk, ragged pixel tiles, and every prologue it takes."""
    import hiputil as hu
    B, HW, W, cout = 3, 150, 15, 128
    x = U(f"pl{cin}.x", (B, HW, cin), -1.5, 1.5)
    w, b = U(f"pl{cin}.w", (cout, cin), -0.3, 0.3), U(f"pl{cin}.b", (cout,))
    wp, xd, bd = hu.pack_pw(ctx, w), hu.dev(x), hu.dev(b)
    assert rel_err(hu.pointwise(ctx, hu.src(xd), wp, bd, B, HW, W, cin, cout).cpu(), F.linear(x, w, b)) < TOL
    # virtual concat whose boundary (c0 = cin/2 + 16 or so) falls inside a 32-channel chunk
    c0 = cin // 2 + (16 if (cin // 2) % 32 == 0 else 0)
    xa, xb = hu.dev(x[..., :c0].contiguous()), hu.dev(x[..., c0:].contiguous())
    assert rel_err(hu.pointwise(ctx, hu.src(xa, xb), wp, bd, B, HW, W, cin, cout).cpu(), F.linear(x, w, b)) < TOL
    # SiLU / LeakyReLU / GroupNorm-affine + SiLU prologues
    assert rel_err(hu.pointwise(ctx, hu.src(xd, None, L.PRO_SILU), wp, bd, B, HW, W, cin, cout).cpu(), F.linear(F.silu(x), w, b)) < TOL
    assert rel_err(hu.pointwise(ctx, hu.src(xd, None, L.PRO_LEAKY), wp, bd, B, HW, W, cin, cout).cpu(),
                   F.linear(F.leaky_relu(x, 0.2), w, b)) < TOL
    mad = U(f"pl{cin}.mad", (B, 3, cin), 0.5, 1.5)
    ref = F.linear(F.silu((x - mad[:, None, 0]) * mad[:, None, 1] + mad[:, None, 2]), w, b)
    assert rel_err(hu.pointwise(ctx, hu.src(xd, None, L.PRO_AFFINE_SILU, mad=hu.dev(mad)), wp, bd, B, HW, W, cin, cout).cpu(), ref) < TOL
    # LayerNorm(x + vec) with row statistics from the pre-pass (the wide-row form; narrow rows without rowstats stay on
    # the single-chunk kernel, covered by test_pointwise_prologues)
    vec, g, be = U(f"pl{cin}.v", (B, cin)), U(f"pl{cin}.g", (cin,), 0.5, 1.5), U(f"pl{cin}.be", (cin,))
    rs, vd = hu.full((B, HW, 2)), hu.dev(vec)
    L.call("nd_layernorm_stats_f32", xd.data_ptr(), cin, vd.data_ptr(), rs.data_ptr(), B, HW, cin, 1e-5, ctx.stream)
    ctx.sync()
    s = hu.src(xd, None, L.PRO_LAYERNORM, vec=vd, gamma=hu.dev(g), beta=hu.dev(be), rowstats=rs)
    ref = F.gelu(F.linear(F.layer_norm(x + vec[:, None], (cin,), g, be, eps=1e-5), w, b))
    assert rel_err(hu.pointwise(ctx, s, wp, bd, B, HW, W, cin, cout, act=L.ACT_GELU).cpu(), ref) < TOL


def test_pointwise_large_tile_kernel(ctx):
    """The one-wave-per-SIMD form of the 1x1 GEMM (pointwise_big_kernel: cin % 64 == 0, cout % 128 == 0 and at least one 128-pixel
    tile per CU): 3 chunks of 64 channels, a ragged last pixel tile, concat on a chunk boundary, every prologue, all epilogue operands."""
    import hiputil as hu
    import functools
    B, HW, W, cin, cout = 2, 8200, 100, 192, 256              # 2 x 65 x 2 = 260 workgroups
    pwf = functools.partial(hu.pointwise, entry="nd_pointwise_gemm_nhwc_f32")
    x = U("big.x", (B, HW, cin), -1.5, 1.5)
    w, b = U("big.w", (cout, cin), -0.2, 0.2), U("big.b", (cout,))
    wp, xd, bd = hu.pack_pw(ctx, w), hu.dev(x), hu.dev(b)
    lin = F.linear(x, w, b)
    assert rel_err(pwf(ctx, hu.src(xd), wp, bd, B, HW, W, cin, cout).cpu(), lin) < TOL
    xa, xb = hu.dev(x[..., :128].contiguous()), hu.dev(x[..., 128:].contiguous())
    assert rel_err(pwf(ctx, hu.src(xa, xb), wp, bd, B, HW, W, cin, cout).cpu(), lin) < TOL
    assert rel_err(pwf(ctx, hu.src(xd, None, L.PRO_SILU), wp, bd, B, HW, W, cin, cout).cpu(), F.linear(F.silu(x), w, b)) < TOL
    assert rel_err(pwf(ctx, hu.src(xd, None, L.PRO_LEAKY), wp, bd, B, HW, W, cin, cout).cpu(), F.linear(F.leaky_relu(x, 0.2), w, b)) < TOL
    mad = U("big.mad", (B, 3, cin), 0.5, 1.5)
    ref = F.linear(F.silu((x - mad[:, None, 0]) * mad[:, None, 1] + mad[:, None, 2]), w, b)
    assert rel_err(pwf(ctx, hu.src(xd, None, L.PRO_AFFINE_SILU, mad=hu.dev(mad)), wp, bd, B, HW, W, cin, cout).cpu(), ref) < TOL
    vec, g, be = U("big.v", (B, cin)), U("big.g", (cin,), 0.5, 1.5), U("big.be", (cin,))
    rs, vd = hu.full((B, HW, 2)), hu.dev(vec)
    L.call("nd_layernorm_stats_f32", xd.data_ptr(), cin, vd.data_ptr(), rs.data_ptr(), B, HW, cin, 1e-5, ctx.stream)
    ctx.sync()
    s = hu.src(xd, None, L.PRO_LAYERNORM, vec=vd, gamma=hu.dev(g), beta=hu.dev(be), rowstats=rs)
    ref = F.gelu(F.linear(F.layer_norm(x + vec[:, None], (cin,), g, be, eps=1e-5), w, b))
    wpl = wp
    assert rel_err(pwf(ctx, s, wpl, bd, B, HW, W, cin, cout, act=L.ACT_GELU).cpu(), ref) < TOL
    # epilogue operands: two residuals, a per-sample vector, the fused ResnetBlock tail silu(GroupNorm-affine(t))
    r0, r1, ov, t = U("big.r0", (B, HW, cout)), U("big.r1", (B, HW, cout)), U("big.ov", (B, cout)), U("big.t", (B, HW, cout), -1.5, 1.5)
    tm = U("big.tm", (B, 3, cout), 0.5, 1.5)
    ref = lin + r0 + r1 + ov[:, None] + F.silu((t - tm[:, None, 0]) * tm[:, None, 1] + tm[:, None, 2])
    out = pwf(ctx, hu.src(xd), wp, bd, B, HW, W, cin, cout, res0=hu.dev(r0), res1=hu.dev(r1), vec=hu.dev(ov), gn_t=hu.dev(t), gn_mad=hu.dev(tm))
    assert rel_err(out.cpu(), ref) < TOL


@pytest.mark.parametrize("cin,cout", [(1024, 2048), (2048, 1024), (1024, 1024)])
def test_pointwise_large_tile_kernel_config4_widths(ctx, cin, cout):
    """pointwise_big_kernel at BASELINE config 4's widths (d=128: FeedForward 1024 -> 2048 -> 1024 and proj_out 1024 -> 1024 on the
    32 x 32 = 1024 tokens of the H/8 stage; 16 / 32 K chunks of 64): LayerNorm prologue + GELU, plain, residual + per-sample vector."""
    import hiputil as hu
    B, HW, W = 4, 1024, 32                                    # 4 x 8 pixel tiles x cout/128 >= 256 workgroups: the large-tile kernel takes it
    bound = 1.0 / np.sqrt(cin)
    x = U(f"c4pw.x.{cin}", (B, HW, cin), -1.5, 1.5)
    w, b = U(f"c4pw.w.{cin}.{cout}", (cout, cin), -bound, bound), U(f"c4pw.b.{cout}", (cout,), -bound, bound)
    wp, xd, bd = hu.pack_pw(ctx, w), hu.dev(x), hu.dev(b)
    lin = F.linear(x, w, b)
    assert rel_err(hu.pointwise(ctx, hu.src(xd), wp, bd, B, HW, W, cin, cout).cpu(), lin) < TOL
    if cin <= 1024:                                           # LayerNorm feeds ff.net.0.0 only (its rows are at most 8 d = 1024 wide)
        vec, g, be = U(f"c4pw.v.{cin}", (B, cin)), U(f"c4pw.g.{cin}", (cin,), 0.5, 1.5), U(f"c4pw.be.{cin}", (cin,))
        rs, vd = hu.full((B, HW, 2)), hu.dev(vec)
        L.call("nd_layernorm_stats_f32", xd.data_ptr(), cin, vd.data_ptr(), rs.data_ptr(), B, HW, cin, 1e-5, ctx.stream)
        ctx.sync()
        s = hu.src(xd, None, L.PRO_LAYERNORM, vec=vd, gamma=hu.dev(g), beta=hu.dev(be), rowstats=rs)
        ref = F.gelu(F.linear(F.layer_norm(x + vec[:, None], (cin,), g, be, eps=1e-5), w, b))
        wpl = hu.pack_pw(ctx, w)
        assert rel_err(hu.pointwise(ctx, s, wpl, bd, B, HW, W, cin, cout, act=L.ACT_GELU).cpu(), ref) < TOL
    r0, ov = U(f"c4pw.r0.{cout}", (B, HW, cout)), U(f"c4pw.ov.{cout}", (B, cout))
    out = hu.pointwise(ctx, hu.src(xd), wp, bd, B, HW, W, cin, cout, res0=hu.dev(r0), vec=hu.dev(ov))
    assert rel_err(out.cpu(), lin + r0 + ov[:, None]) < TOL


def test_affine_silu_add_and_rmsnorm(ctx):
    import hiputil as hu
    B, HW, Cc = 3, 100, 48
    t, r0, r1 = U("as.t", (B, HW, Cc), -2, 2), U("as.r0", (B, HW, Cc)), U("as.r1", (B, HW, Cc))
    mad = U("as.mad", (B, 3, Cc), 0.5, 1.5)
    out = torch.empty(B, HW, Cc, device=hu.DEV)
    td, md, r0d, r1d = hu.dev(t), hu.dev(mad), hu.dev(r0), hu.dev(r1)
    L.call("nd_affine_silu_add_f32", td.data_ptr(), Cc, md.data_ptr(), r0d.data_ptr(), Cc, r1d.data_ptr(), Cc, out.data_ptr(), Cc, B, HW, Cc, ctx.stream)
    ctx.sync()
    ref = F.silu((t - mad[:, None, 0]) * mad[:, None, 1] + mad[:, None, 2]) + r0 + r1
    assert rel_err(out.cpu(), ref) < TOL
    g = U("rms.g", (Cc,), 0.5, 1.5)
    gd = hu.dev(g)
    L.call("nd_rmsnorm_nhwc_f32", td.data_ptr(), Cc, gd.data_ptr(), out.data_ptr(), Cc, B, HW, Cc, ctx.stream)
    ctx.sync()
    ref = O.rms_norm(g.view(1, Cc, 1, 1), t.permute(0, 2, 1).reshape(B, Cc, HW, 1)).reshape(B, Cc, HW).permute(0, 2, 1)
    assert rel_err(out.cpu(), ref) < TOL


def test_small_ops(ctx):
    import hiputil as hu
    B, d = 5, 64
    # sinusoidal embedding + time MLP rows
    time = torch.tensor([0, 1, 500, 998, 999])
    half = d // 2
    freqs = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1))).float()
    emb = torch.empty(B, d, device=hu.DEV)
    td, fd = hu.dev(time), hu.dev(freqs)
    L.call("nd_sinusoidal_time_emb_f32", td.data_ptr(), fd.data_ptr(), emb.data_ptr(), B, half, ctx.stream)
    ctx.sync()
    assert rel_err(emb.cpu(), O.sinusoidal_pos_emb(time, d)) < 1e-5
    K, N = 256, 333
    x, w, b = U("lr.x", (B, K), -2, 2), U("lr.w", (N, K), -0.2, 0.2), U("lr.b", (N,))
    out = torch.empty(B, N, device=hu.DEV)
    xd, wd, bd = hu.dev(x), hu.dev(w), hu.dev(b)
    L.call("nd_linear_rows_f32", xd.data_ptr(), K, wd.data_ptr(), bd.data_ptr(), out.data_ptr(), N, B, K, N, L.ACT_SILU, L.ACT_GELU, ctx.stream)
    ctx.sync()
    assert rel_err(out.cpu(), F.gelu(F.linear(F.silu(x), w, b))) < TOL
    # embedding
    table, idx = U("em.t", (100, 16)), torch.tensor([0, 74, 99, 3, 3])
    o = torch.empty(B, 16, device=hu.DEV)
    tabd, idxd = hu.dev(table), hu.dev(idx)
    L.call("nd_embedding_rows_f32", idxd.data_ptr(), tabd.data_ptr(), o.data_ptr(), B, 100, 16, ctx.stream)
    ctx.sync()
    assert torch.equal(o.cpu(), table[idx])
    # pos_enc
    pos = synth.make_position(2, 24, seed=3)
    w2, b2 = U("pe.w", (8, 2, 1, 1)), U("pe.b", (8,))
    ref = O.learned_sinusoidal_pos_emb({"p.weights.weight": w2, "p.weights.bias": b2}, "p", pos)
    o = torch.empty(2, 24, 24, 24, device=hu.DEV)
    pd, w2d, b2d = hu.dev(pos), hu.dev(w2), hu.dev(b2)
    L.call("nd_pos_enc_f32", pd.data_ptr(), w2d.data_ptr(), b2d.data_ptr(), o.data_ptr(), 2, 24, 24, 8, ctx.stream)
    ctx.sync()
    assert rel_err(hu.nchw(o), ref) < 1e-5
    # layout round trip
    x4 = U("lay.x", (3, 4, 8, 16))
    a, bb = torch.empty(3, 8, 16, 4, device=hu.DEV), torch.empty(3, 4, 8, 16, device=hu.DEV)
    x4d = hu.dev(x4)
    L.call("nd_nchw_to_nhwc_f32", x4d.data_ptr(), a.data_ptr(), 3, 4, 8, 16, ctx.stream)
    L.call("nd_nhwc_to_nchw_f32", a.data_ptr(), bb.data_ptr(), 3, 4, 8, 16, ctx.stream)
    ctx.sync()
    assert torch.equal(a.cpu(), x4.permute(0, 2, 3, 1)) and torch.equal(bb.cpu(), x4)


@pytest.mark.parametrize("shape", [(2, 16, 16, 16), (1, 40, 24, 48), (2, 64, 64, 64), (1, 19, 45, 128), (1, 16, 32, 20)])
def test_conv7x7(ctx, shape):
    """init_conv (Diffusion_arch.py:478)."""
    import hiputil as hu
    B, H, W, cout = shape
    x, w, b = U("c7.x", (B, 4, H, W), -1.5, 1.5), U("c7.w", (cout, 4, 7, 7), -0.1, 0.1), U("c7.b", (cout,))
    wp, out = torch.empty(196 * cout, device=hu.DEV), torch.empty(B, H, W, cout, device=hu.DEV)
    xd, wd, bd = hu.nhwc(x), hu.dev(w), hu.dev(b)
    L.call("nd_pack_conv7x7_weight", wd.data_ptr(), wp.data_ptr(), cout, ctx.stream)
    L.call("nd_conv7x7_c4_f32", xd.data_ptr(), wp.data_ptr(), bd.data_ptr(), out.data_ptr(), cout, B, H, W, cout, ctx.stream)
    ctx.sync()
    ref = F.conv2d(x, w, b, padding=3)
    assert rel_err(hu.nchw(out), ref) < TOL
    # the split-product form (r6): three bf16 terms per operand on v_mfma_f32_32x32x16_bf16, same tolerance, no worse than 1.5 x the fp32 kernel against fp64
    ws, out2 = hu.full((ctx.lib.nd_pack_conv7x7_weight_split_floats(cout),)), hu.full((B, H, W, cout))
    L.call("nd_pack_conv7x7_weight_split", wd.data_ptr(), ws.data_ptr(), cout, ctx.stream)
    L.call("nd_conv7x7_c4_split_f32", xd.data_ptr(), ws.data_ptr(), bd.data_ptr(), out2.data_ptr(), cout, B, H, W, cout, ctx.stream)
    ctx.sync()
    assert rel_err(hu.nchw(out2), ref) < TOL
    ref64 = F.conv2d(x.double(), w.double(), b.double(), padding=3)
    e32, esp = (hu.nchw(out).double() - ref64).abs().max().item(), (hu.nchw(out2).double() - ref64).abs().max().item()
    assert esp <= 1.5 * e32 + 1e-7, (esp, e32)
    first = out2.clone()
    torch.cuda.synchronize()
    L.call("nd_conv7x7_c4_split_f32", xd.data_ptr(), ws.data_ptr(), bd.data_ptr(), out2.data_ptr(), cout, B, H, W, cout, ctx.stream)
    ctx.sync()
    assert torch.equal(out2, first)


@pytest.mark.parametrize("N", [64, 1024, 200])
def test_attention_mfma(ctx, N):
    import hiputil as hu
    B, heads, dh = 2, 4, 32
    qkv = U(f"att.{N}", (B, N, 3 * heads * dh), -2, 2)
    out = hu.full((B, N, heads * dh))
    qd = hu.dev(qkv)
    L.call("nd_attention_mfma_f32", qd.data_ptr(), 3 * heads * dh, out.data_ptr(), heads * dh, B, N, heads, dh, ctx.stream)
    ctx.sync()
    q, k, v = (t.reshape(B, N, heads, dh).permute(0, 2, 1, 3) for t in qkv.chunk(3, dim=-1))
    ref = torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5, -1) @ v
    assert rel_err(out.cpu(), ref.permute(0, 2, 1, 3).reshape(B, N, heads * dh)) < TOL


def test_philox_matches_oracle_and_is_shard_invariant(ctx):
    import hiputil as hu
    B, HW, Cc, seed = 3, 64, 4, 0x1234567890ABCDEF
    out = torch.empty(B, HW, Cc, device=hu.DEV)
    L.call("nd_philox_normal_f32", out.data_ptr(), C.c_uint64(seed), 5, 7, B, HW, Cc, ctx.stream)
    ctx.sync()
    q = np.arange(HW, dtype=np.uint32)
    for b in range(B):
        ctr = np.stack([q, np.full(HW, 5 + b, np.uint32), np.full(HW, 8, np.uint32), np.zeros(HW, np.uint32)], -1)
        key = np.tile(np.array([[seed & 0xFFFFFFFF, seed >> 32]], dtype=np.uint32), (HW, 1))
        ref = O.philox_normal4(O.philox4x32_10(ctr, key))
        np.testing.assert_allclose(out[b].cpu().numpy(), ref, atol=2e-5, rtol=1e-4)
    # rows of a shard starting at sample 6 == rows 1.. of the shard starting at 5
    out2 = torch.empty(2, HW, Cc, device=hu.DEV)
    L.call("nd_philox_normal_f32", out2.data_ptr(), C.c_uint64(seed), 6, 7, 2, HW, Cc, ctx.stream)
    ctx.sync()
    assert torch.equal(out2.cpu(), out[1:].cpu())
    big = torch.empty(64, 4096, 4, device=hu.DEV)
    L.call("nd_philox_normal_f32", big.data_ptr(), C.c_uint64(1), 0, -1, 64, 4096, 4, ctx.stream)
    ctx.sync()
    assert abs(float(big.mean())) < 5e-3 and abs(float(big.std()) - 1.0) < 5e-3


@pytest.mark.parametrize("N", [64, 2048 + 200])
def test_linear_attention_matches_reference_semantics(ctx, N, golden):
    """LinearAttention core (Diffusion_arch.py:223-234) vs the oracle's einsum restatement; plus the whole block
    (RMSNorm -> qkv -> core -> to_out -> RMSNorm) against the golden captured from the reference class."""
    import hiputil as hu
    B, heads, dh = 2, 4, 32
    qkv = U(f"lat.{N}", (B, N, 3 * heads * dh), -2, 2)
    out = hu.full((B, N, heads * dh))
    ws = hu.full((ctx.lib.nd_linear_attention_workspace_floats(B, N, heads),))
    qd = hu.dev(qkv)
    L.call("nd_linear_attention_f32", qd.data_ptr(), 3 * heads * dh, out.data_ptr(), heads * dh, ws.data_ptr(), B, N, heads, dh, ctx.stream)
    ctx.sync()
    q, k, v = (t.reshape(B, N, heads, dh).permute(0, 2, 3, 1) for t in qkv.chunk(3, dim=-1))      # b h c n
    q = q.softmax(dim=-2) * dh ** -0.5
    k = k.softmax(dim=-1)
    ref = torch.einsum("bhde,bhdn->bhen", torch.einsum("bhdn,bhen->bhde", k, v), q)                # b h e n
    assert rel_err(out.cpu(), ref.permute(0, 3, 1, 2).reshape(B, N, heads * dh)) < TOL
    if N != 64:
        return
    # full block on the golden input (B=2, C=128, 8x8)
    from noisediff_amd.spec import attention_param_spec
    Cc = 128
    xa = synth.uniform(7, "mod.xa", (B, Cc, 8, 8), -1.5, 1.5)
    sda = synth.make_state_dict(attention_param_spec("mid_attn", Cc), 0)
    g2 = synth.uniform(7, "mod.lat_g", (1, Cc, 1, 1), 0.5, 1.5)
    x = hu.nhwc(xa).view(B, 64, Cc)
    xn, y, yn = hu.full((B, 64, Cc)), hu.full((B, 64, Cc)), hu.full((B, 64, Cc))
    ones, g2d = hu.dev(torch.ones(Cc)), hu.dev(g2.reshape(-1))
    L.call("nd_rmsnorm_nhwc_f32", x.data_ptr(), Cc, ones.data_ptr(), xn.data_ptr(), Cc, B, 64, Cc, ctx.stream)
    ctx.sync()
    qkv2 = hu.pointwise(ctx, hu.src(xn), hu.pack_pw(ctx, sda["mid_attn.to_qkv.weight"]), None, B, 64, 8, Cc, 384)
    att = hu.full((B, 64, 128))
    L.call("nd_linear_attention_f32", qkv2.data_ptr(), 384, att.data_ptr(), 128, ws.data_ptr(), B, 64, heads, dh, ctx.stream)
    ctx.sync()
    y = hu.pointwise(ctx, hu.src(att), hu.pack_pw(ctx, sda["mid_attn.to_out.weight"]), hu.dev(sda["mid_attn.to_out.bias"]), B, 64, 8, 128, Cc)
    L.call("nd_rmsnorm_nhwc_f32", y.data_ptr(), Cc, g2d.data_ptr(), yn.data_ptr(), Cc, B, 64, Cc, ctx.stream)
    ctx.sync()
    got = yn.view(B, 8, 8, Cc).permute(0, 3, 1, 2).cpu().numpy()
    assert rel_err(got, golden("modules", "mod.linear_attention")) < 1e-4


WINO_CASES = {  # (B, H, W, cin, cout)
    "concat_uneven": (2, 32, 48, 96, 32),         # 64 + 32 channel sources with different pixel strides
    "square": (2, 32, 32, 32, 64),
    "ragged_edges": (2, 24, 40, 16, 48),          # partial tiles in x and y, partial K chunk, cout mask
    "wide_k": (1, 16, 16, 96, 128),               # 3 K chunks, two N tiles
    "big": (4, 64, 64, 64, 64),
}


WINO_ENTRIES = ["nd_conv3x3_wino_nhwc_f32", "nd_conv3x3_wino2_nhwc_f32"]


@pytest.mark.parametrize("entry", WINO_ENTRIES)
@pytest.mark.parametrize("case", sorted(WINO_CASES))
def test_conv3x3_winograd_matches_direct_semantics(ctx, case, entry):
    """Winograd F(2x2,3x3) kernel == nn.Conv2d(3, padding=1) (tolerance 1e-5: fp32 transforms), stats included."""
    import hiputil as hu
    B, H, W, cin, cout = WINO_CASES[case]
    x = U(case + ".wx", (B, cin, H, W), -1.5, 1.5)
    w = U(case + ".ww", (cout, cin, 3, 3), -0.2, 0.2)
    b = U(case + ".wb", (cout,))
    ref = F.conv2d(x, w, b, padding=1)
    wd = hu.dev(w)
    wp = hu.full((ctx.lib.nd_pack_conv3x3_wino_weight_floats(cin, cout),))
    L.call("nd_pack_conv3x3_wino_weight", wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream)
    ctx.sync()

    def run(s):
        out = hu.full((B, H, W, cout))
        slots = ctx.lib.nd_conv3x3_wino_stat_slots(H, W)
        st, sc = hu.full((B, slots, cout, 2)), hu.full((slots,))
        d = L.Conv3x3()
        d.src, d.weight, d.bias, d.out, d.stats, d.slot_count = s, wp.data_ptr(), bd.data_ptr(), out.data_ptr(), st.data_ptr(), sc.data_ptr()
        d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
        L.call(entry, C.byref(d), ctx.stream)
        ctx.sync()
        return out, st, sc, slots

    bd = hu.dev(b)
    out, st, sc, slots = run(hu.src(hu.nhwc(x)))
    assert rel_err(hu.nchw(out), ref) < 1e-5
    groups = 8 if cout % 8 == 0 else 2
    gamma, beta = U(case + ".wg", (cout,), 0.5, 1.5), U(case + ".wbe", (cout,))
    mad = hu.gn_finalize(ctx, st, sc, slots, hu.dev(gamma), hu.dev(beta), None, B, cout, groups).cpu()
    gn = F.group_norm(ref, groups, gamma, beta, eps=1e-5)
    mine = (ref - mad[:, 0, :, None, None]) * mad[:, 1, :, None, None] + mad[:, 2, :, None, None]
    assert rel_err(mine, gn) < 1e-5
    # fused prologue (GroupNorm-affine + SiLU, zero padding after the activation) and virtual concat
    M, A, D = U(case + ".wM", (B, cin)), U(case + ".wA", (B, cin), 0.5, 1.5), U(case + ".wD", (B, cin))
    act = F.silu((x - M[:, :, None, None]) * A[:, :, None, None] + D[:, :, None, None])
    out, *_ = run(hu.src(hu.nhwc(x), None, L.PRO_AFFINE_SILU, mad=hu.dev(torch.stack((M, A, D), 1))))
    assert rel_err(hu.nchw(out), F.conv2d(act, w, b, padding=1)) < 1e-5
    c0 = cin // 2 // 4 * 4
    if entry == "nd_conv3x3_wino2_nhwc_f32" and c0 % 32:
        # a 32-channel K chunk must not straddle the concat sources there: refused loudly, the engine routes such layers
        # to nd_conv3x3_wino_nhwc_f32; where it fits, split on the chunk boundary instead
        with pytest.raises(L.HipError, match="straddle"):
            run(hu.src(hu.nhwc(x[:, :c0]), hu.nhwc(x[:, c0:])))
        if cin <= 32:
            return
        c0 = 32
    out, *_ = run(hu.src(hu.nhwc(x[:, :c0]), hu.nhwc(x[:, c0:])))
    assert rel_err(hu.nchw(out), ref) < 1e-5


def _run_wino4(ctx, s, wp, bd, B, H, W, cin, cout, stats=True, entry="nd_conv3x3_wino4_nhwc_f32"):
    import hiputil as hu
    out = hu.full((B, H, W, cout))
    d = L.Conv3x3()
    d.src, d.weight, d.bias, d.out = s, wp.data_ptr(), bd.data_ptr(), out.data_ptr()
    d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
    st = sc = None
    slots = ctx.lib.nd_conv3x3_wino4_stat_slots(H, W)            # one per 16 x 16 tile (the F(2x2) kernels: two)
    if stats:
        st, sc = hu.full((B, slots, cout, 2)), hu.full((slots,))
        d.stats, d.slot_count = st.data_ptr(), sc.data_ptr()
    L.call(entry, C.byref(d), ctx.stream)
    ctx.sync()
    return out, st, sc, slots


def _check_gn(ctx, case, ref, st, sc, slots, B, cout, tol):
    import hiputil as hu
    groups = 8 if cout % 8 == 0 else 2
    gamma, beta = U(case + ".wg", (cout,), 0.5, 1.5), U(case + ".wbe", (cout,))
    mad = hu.gn_finalize(ctx, st, sc, slots, hu.dev(gamma), hu.dev(beta), None, B, cout, groups).cpu()
    mine = (ref - mad[:, 0, :, None, None]) * mad[:, 1, :, None, None] + mad[:, 2, :, None, None]
    assert rel_err(mine, F.group_norm(ref, groups, gamma, beta, eps=1e-5)) < tol


@pytest.mark.parametrize("case", sorted(WINO_CASES))
def test_conv3x3_winograd_f4x4_matches_direct_semantics(ctx, case):
    """The F(4x4,3x3) kernel (conv3x3_wino4.hip) == nn.Conv2d(3, padding=1), GroupNorm partials included.
    Tolerance 5e-5: its fp32 transforms carry entries up to 8 and 1/24 (F(2x2,3x3): 1e-5)."""
    import hiputil as hu
    B, H, W, cin, cout = WINO_CASES[case]
    cin = max(cin, 24)                                  # at least two 16-channel K chunks (ragged_edges: 16 -> 24, the second one partial)
    x = U(case + ".wx", (B, cin, H, W), -1.5, 1.5)
    w = U(case + ".ww", (cout, cin, 3, 3), -0.2, 0.2)
    b = U(case + ".wb", (cout,))
    ref = F.conv2d(x, w, b, padding=1)
    wd, bd = hu.dev(w), hu.dev(b)
    wp = hu.full((ctx.lib.nd_pack_conv3x3_wino4_weight_floats(cin, cout),))
    L.call("nd_pack_conv3x3_wino4_weight", wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream)
    ctx.sync()
    run = lambda s, stats=True: _run_wino4(ctx, s, wp, bd, B, H, W, cin, cout, stats)

    out, st, sc, slots = run(hu.src(hu.nhwc(x)))
    assert rel_err(hu.nchw(out), ref) < 5e-5
    _check_gn(ctx, case, ref, st, sc, slots, B, cout, 5e-5)
    M, A, D = U(case + ".wM", (B, cin)), U(case + ".wA", (B, cin), 0.5, 1.5), U(case + ".wD", (B, cin))
    act = F.silu((x - M[:, :, None, None]) * A[:, :, None, None] + D[:, :, None, None])
    out, *_ = run(hu.src(hu.nhwc(x), None, L.PRO_AFFINE_SILU, mad=hu.dev(torch.stack((M, A, D), 1))), stats=False)
    assert rel_err(hu.nchw(out), F.conv2d(act, w, b, padding=1)) < 5e-5
    c0 = cin // 2 // 16 * 16                                             # virtual concat: a 16-channel chunk must not straddle the sources
    if c0:
        out, *_ = run(hu.src(hu.nhwc(x[:, :c0]), hu.nhwc(x[:, c0:])))
        assert rel_err(hu.nchw(out), ref) < 5e-5
    else:
        with pytest.raises(L.HipError, match="straddle"):
            run(hu.src(hu.nhwc(x[:, :8]), hu.nhwc(x[:, 8:])))
    # the per-pixel scale / shift map of ResnetBlock2 and LSID's LeakyReLU prologues (whole input / second concat source only)
    sc, sh = U(case + ".sc", (B, cin, H, W), -0.5, 0.5), U(case + ".sh", (B, cin, H, W), -0.5, 0.5)
    actm = F.silu(((x - M[:, :, None, None]) * A[:, :, None, None] + D[:, :, None, None]) * (sc + 1) + sh)
    sm = hu.src(hu.nhwc(x), None, L.PRO_AFFINE_MAP_SILU, mad=hu.dev(torch.stack((M, A, D), 1)), map=hu.nhwc(torch.cat((sc, sh), 1)))
    out, *_ = run(sm, stats=False)
    assert rel_err(hu.nchw(out), F.conv2d(actm, w, b, padding=1)) < 5e-5
    if cin % 16 == 0:                                                    # the 16-channel-blocked map layout (nd_src.map_blocked): [chunk][scale 16 | shift 16]
        blk = torch.stack((sc.permute(0, 2, 3, 1).reshape(B, H, W, cin // 16, 16), sh.permute(0, 2, 3, 1).reshape(B, H, W, cin // 16, 16)), 4).reshape(B, H, W, 2 * cin)
        smb = hu.src(hu.nhwc(x), None, L.PRO_AFFINE_MAP_SILU, mad=hu.dev(torch.stack((M, A, D), 1)), map=hu.dev(blk), map_blocked=1)
        out, *_ = run(smb, stats=False)
        assert rel_err(hu.nchw(out), F.conv2d(actm, w, b, padding=1)) < 5e-5
        for other in ("nd_conv3x3_wino2_nhwc_f32", "nd_conv3x3_nhwc_f32"):   # the other kernels read the planar layout only: refused loudly
            d = L.Conv3x3()
            d.src, d.weight, d.bias, d.out = smb, wp.data_ptr(), bd.data_ptr(), out.data_ptr()
            d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
            assert getattr(ctx.lib, other)(C.byref(d), ctx.stream) != 0
    # ND_PRO_AFFINE_GENMAP_SILU: the same modulation with the maps formed in the kernel -- scale | shift = mlp[1] (1x1, 8 -> 2 cin) of the activated position
    # embedding (Diffusion_arch.py:177,188) on the matrix pipe, from 32 bytes per pixel
    pe = U(case + ".pe", (B, 8, H, W), -2.0, 2.0)
    gw, gb = U(case + ".gw", (2 * cin, 8), -0.4, 0.4), U(case + ".gb", (2 * cin,), -0.3, 0.3)
    gen = hu.src(hu.nhwc(x), None, L.PRO_AFFINE_GENMAP_SILU, mad=hu.dev(torch.stack((M, A, D), 1)), map=hu.nhwc(pe), gamma=hu.dev(gw), beta=hu.dev(gb))
    if cin % 16 == 0:
        ss = F.conv2d(pe.double(), gw.double()[:, :, None, None], gb.double()).float()
        actg = F.silu(((x - M[:, :, None, None]) * A[:, :, None, None] + D[:, :, None, None]) * (ss[:, :cin] + 1) + ss[:, cin:])
        out, st, sc_, slots = run(gen)
        assert rel_err(hu.nchw(out), F.conv2d(actg, w, b, padding=1)) < 5e-5
        _check_gn(ctx, case + ".gen", F.conv2d(actg, w, b, padding=1), st, sc_, slots, B, cout, 5e-5)
    else:
        with pytest.raises(L.HipError, match="in-kernel map prologue"):
            run(gen, stats=False)
    out, *_ = run(hu.src(hu.nhwc(x), None, L.PRO_LEAKY), stats=False)
    assert rel_err(hu.nchw(out), F.conv2d(F.leaky_relu(x, 0.2), w, b, padding=1)) < 5e-5
    if c0:
        out, *_ = run(hu.src(hu.nhwc(x[:, :c0]), hu.nhwc(x[:, c0:]), L.PRO_LEAKY_SECOND), stats=False)
        assert rel_err(hu.nchw(out), F.conv2d(torch.cat((x[:, :c0], F.leaky_relu(x[:, c0:], 0.2)), 1), w, b, padding=1)) < 5e-5
    if H % 2 == 0 and W % 2 == 0:                                        # nearest-x2 upsample addressing (Upsample's conv, Diffusion_arch.py:74-75)
        xs = x[:, :, : H // 2, : W // 2].contiguous()
        out, *_ = run(hu.src(hu.nhwc(xs), upsample=1), stats=False)
        assert rel_err(hu.nchw(out), F.conv2d(F.interpolate(xs, scale_factor=2, mode="nearest"), w, b, padding=1)) < 5e-5


@pytest.mark.parametrize("shape", [(4, 256, 256, 64, 64), (3, 80, 112, 32, 48), (2, 64, 96, 128, 64)])
def test_conv3x3_winograd_f4x4_maps_formed_in_the_kernel_equal_the_maps_read(ctx, shape):
    """ND_PRO_AFFINE_GENMAP_SILU against ND_PRO_AFFINE_MAP_SILU on the maps torch computes from the same operands (mlp[1] of the activated position embedding) and
    against torch's convolution: several tiles per workgroup (the position embedding is loaded once per TILE and kept across its chunks), two to eight chunks,
    border regions, a cout mask.  The two prologues differ in the summation order of the eight map terms only."""
    import hiputil as hu
    B, H, W, cin, cout = shape
    x = U("gen.x", (B, cin, H, W), -1.5, 1.5)
    w, b = U("gen.w", (cout, cin, 3, 3), -0.2, 0.2), U("gen.b", (cout,))
    M, A, D = U("gen.M", (B, cin)), U("gen.A", (B, cin), 0.5, 1.5), U("gen.D", (B, cin))
    pe = U("gen.pe", (B, 8, H, W), -2.0, 2.0)
    gw, gb = U("gen.gw", (2 * cin, 8), -0.4, 0.4), U("gen.gb", (2 * cin,), -0.3, 0.3)
    ss = F.conv2d(pe, gw[:, :, None, None], gb)
    wd, bd = hu.dev(w), hu.dev(b)
    wp = hu.full((ctx.lib.nd_pack_conv3x3_wino4_weight_floats(cin, cout),))
    L.call("nd_pack_conv3x3_wino4_weight", wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream)
    ctx.sync()
    mad = hu.dev(torch.stack((M, A, D), 1))
    got, st, sc, slots = _run_wino4(ctx, hu.src(hu.nhwc(x), None, L.PRO_AFFINE_GENMAP_SILU, mad=mad, map=hu.nhwc(pe), gamma=hu.dev(gw), beta=hu.dev(gb)), wp, bd, B, H, W, cin, cout)
    read, *_ = _run_wino4(ctx, hu.src(hu.nhwc(x), None, L.PRO_AFFINE_MAP_SILU, mad=mad, map=hu.nhwc(ss)), wp, bd, B, H, W, cin, cout)
    act = F.silu(((x - M[:, :, None, None]) * A[:, :, None, None] + D[:, :, None, None]) * (ss[:, :cin] + 1) + ss[:, cin:])
    ref = F.conv2d(act, w, b, padding=1)
    assert rel_err(hu.nchw(got), ref) < 5e-5
    assert rel_err(hu.nchw(got), hu.nchw(read)) < 1e-5
    _check_gn(ctx, "gen", ref, st, sc, slots, B, cout, 5e-5)


def test_conv3x3_winograd_f4x4_channel_slice_of_a_wider_tensor(ctx):
    """Sources with a pixel stride above their channel count (a channel slice of a wider NHWC tensor; stride 68 floats: the
    border padding must not depend on the stride), images with every kind of border region, second region row / batch offsets."""
    import hiputil as hu
    B, H, W, cin, cout, ld = 2, 48, 96, 32, 64, 68
    x = U("slice.x", (B, cin, H, W), -1.5, 1.5)
    w = U("slice.w", (cout, cin, 3, 3), -0.2, 0.2)
    b = U("slice.b", (cout,))
    wide = torch.full((B, H, W, ld), float("nan"))
    wide[..., 20:20 + cin] = x.permute(0, 2, 3, 1)
    wide_d = hu.dev(wide)
    wd, bd = hu.dev(w), hu.dev(b)
    wp = hu.full((ctx.lib.nd_pack_conv3x3_wino4_weight_floats(cin, cout),))
    L.call("nd_pack_conv3x3_wino4_weight", wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream)
    ctx.sync()
    s = L.Src()
    s.p0, s.c0, s.ld0, s.mode = wide_d.data_ptr() + 20 * 4, cin, ld, L.PRO_NONE
    s._refs = [wide_d]
    out, *_ = _run_wino4(ctx, s, wp, bd, B, H, W, cin, cout, stats=False)
    assert rel_err(hu.nchw(out), F.conv2d(x, w, b, padding=1)) < 5e-5


# The shapes that carry the bench workload (SURVEY Appendix A at d=64, 256x256: H/8 = 32x32 with 256..768 input channels and
# 8 output tiles, 16..24 K chunks; H/4 concat; the full-resolution upsample conv; the 64 -> 64 layer at 256x256).
# (B, H, W, cin, cout, first-source channels of a virtual concat or 0, nearest-x2 upsample addressing)
HEADLINE_CASES = {
    "h8_512_512": (2, 32, 32, 512, 512, 0, 0),
    "h8_768cat_512": (2, 32, 32, 768, 512, 512, 0),
    "h8_256_512": (1, 32, 32, 256, 512, 0, 0),
    "h4_384cat_256": (2, 64, 64, 384, 256, 256, 0),
    "h1_up_128_64": (2, 128, 128, 128, 64, 0, 1),
    "h1_64_64": (1, 256, 256, 64, 64, 0, 0),
    # BASELINE config 4 (d=128, 256x256): the H/8 stage at 32x32 with 1024 / 1536 (= 1024 + 512 concat) input channels -- 64 / 96 K chunks of
    # the F(4x4) kernel, 16 cout tiles -- and the 1024 -> 512 convs behind it (plain at 32x32, with nearest-x2 addressing at 64x64)
    "c4_1024_1024": (1, 32, 32, 1024, 1024, 0, 0),
    "c4_1536cat_1024": (1, 32, 32, 1536, 1024, 1024, 0),
    "c4_1024_512": (1, 32, 32, 1024, 512, 0, 0),
    "c4_up_1024_512": (1, 64, 64, 1024, 512, 0, 1),
}


def _headline_inputs(case):
    B, H, W, cin, cout, c0, up = HEADLINE_CASES[case]
    hs, ws = (H // 2, W // 2) if up else (H, W)
    bound = 1.0 / np.sqrt(9 * cin)                                   # PyTorch default init: outputs stay O(1)
    x = U(case + ".x", (B, cin, hs, ws), -1.5, 1.5)
    w = U(case + ".w", (cout, cin, 3, 3), -bound, bound)
    b = U(case + ".b", (cout,), -bound, bound)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    return x, xin, w, b


@pytest.mark.parametrize("case", sorted(HEADLINE_CASES))
def test_conv3x3_wino2_headline_shapes(ctx, case):
    """wino2 at the bench workload's own layer shapes: output, GroupNorm partials, affine + SiLU prologue, bitwise repeat."""
    import hiputil as hu
    B, H, W, cin, cout, c0, up = HEADLINE_CASES[case]
    x, xin, w, b = _headline_inputs(case)
    ref = F.conv2d(xin, w, b, padding=1)
    wd, bd = hu.dev(w), hu.dev(b)
    wp = hu.full((ctx.lib.nd_pack_conv3x3_wino_weight_floats(cin, cout),))
    L.call("nd_pack_conv3x3_wino_weight", wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream)
    ctx.sync()
    slots = ctx.lib.nd_conv3x3_wino_stat_slots(H, W)

    def run(s):
        out, st, sc = hu.full((B, H, W, cout)), hu.full((B, slots, cout, 2)), hu.full((slots,))
        d = L.Conv3x3()
        d.src, d.weight, d.bias, d.out, d.stats, d.slot_count = s, wp.data_ptr(), bd.data_ptr(), out.data_ptr(), st.data_ptr(), sc.data_ptr()
        d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
        L.call("nd_conv3x3_wino2_nhwc_f32", C.byref(d), ctx.stream)
        ctx.sync()
        return out, st, sc

    s = hu.src(hu.nhwc(x[:, :c0]), hu.nhwc(x[:, c0:])) if c0 else hu.src(hu.nhwc(x), upsample=up)
    out, st, sc = run(s)
    assert rel_err(hu.nchw(out), ref) < 1e-5
    gamma, beta = U(case + ".g", (cout,), 0.5, 1.5), U(case + ".be", (cout,))
    mad = hu.gn_finalize(ctx, st, sc, slots, hu.dev(gamma), hu.dev(beta), None, B, cout, 8).cpu()
    mine = (ref - mad[:, 0, :, None, None]) * mad[:, 1, :, None, None] + mad[:, 2, :, None, None]
    assert rel_err(mine, F.group_norm(ref, 8, gamma, beta, eps=1e-5)) < 1e-5
    out2, st2, _ = run(s)                                               # inline-asm MFMAs: a missed wait state shows as run-to-run noise
    assert torch.equal(out.cpu(), out2.cpu()) and torch.equal(st.cpu(), st2.cpu())
    if not up:
        M, A, D = U(case + ".M", (B, cin)), U(case + ".A", (B, cin), 0.5, 1.5), U(case + ".D", (B, cin))
        act = F.silu((x - M[:, :, None, None]) * A[:, :, None, None] + D[:, :, None, None])
        out, *_ = run(hu.src(hu.nhwc(x), None, L.PRO_AFFINE_SILU, mad=hu.dev(torch.stack((M, A, D), 1))))
        assert rel_err(hu.nchw(out), F.conv2d(act, w, b, padding=1)) < 1e-5


@pytest.mark.parametrize("case", sorted(HEADLINE_CASES))
def test_conv3x3_wino4_headline_shapes(ctx, case):
    """F(4x4,3x3) at the same shapes (5e-5: its transforms carry entries up to 8 and 1/24): output, GroupNorm partials, bitwise repeat,
    affine + SiLU prologue."""
    import hiputil as hu
    B, H, W, cin, cout, c0, up = HEADLINE_CASES[case]
    x, xin, w, b = _headline_inputs(case)
    ref = F.conv2d(xin, w, b, padding=1)
    wd, bd = hu.dev(w), hu.dev(b)
    wp = hu.full((ctx.lib.nd_pack_conv3x3_wino4_weight_floats(cin, cout),))
    L.call("nd_pack_conv3x3_wino4_weight", wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream)
    ctx.sync()
    run = lambda s: _run_wino4(ctx, s, wp, bd, B, H, W, cin, cout)
    s = hu.src(hu.nhwc(x[:, :c0]), hu.nhwc(x[:, c0:])) if c0 else hu.src(hu.nhwc(x), upsample=up)
    out, st, sc, slots = run(s)
    assert rel_err(hu.nchw(out), ref) < 5e-5
    _check_gn(ctx, case, ref, st, sc, slots, B, cout, 5e-5)
    out2, st2, *_ = run(s)
    assert torch.equal(out.cpu(), out2.cpu()) and torch.equal(st.cpu(), st2.cpu())
    if not up:
        M, A, D = U(case + ".M", (B, cin)), U(case + ".A", (B, cin), 0.5, 1.5), U(case + ".D", (B, cin))
        act = F.silu((x - M[:, :, None, None]) * A[:, :, None, None] + D[:, :, None, None])
        out, *_ = run(hu.src(hu.nhwc(x), None, L.PRO_AFFINE_SILU, mad=hu.dev(torch.stack((M, A, D), 1))))
        assert rel_err(hu.nchw(out), F.conv2d(act, w, b, padding=1)) < 5e-5


# BASELINE config 2 (d=64, 128x128): the H/8 stage is 16 x 16 pixels -- narrower than the 16 x 32 regions of the one-workgroup form -- and the H/4 stage
# has eight of those regions per sample (SURVEY Appendix A, H/8 and H/4 columns; Diffusion_arch.py:533,547).  (B, H, W, cin, cout, concat c0, upsample)
CFG2_CASES = {
    "c2_h8_512_512": (16, 16, 16, 512, 512, 0, 0),
    "c2_h8_768cat_512": (16, 16, 16, 768, 512, 512, 0),
    "c2_h8_256_512": (16, 16, 16, 256, 512, 0, 0),
    "c2_h4_384cat_256": (16, 32, 32, 384, 256, 256, 0),
    "c2_h4_up_512_256": (2, 32, 32, 512, 256, 0, 1),
}
W16_CASES = {**{k: v + (0, 0) for k, v in WINO_CASES.items()}, **{k: HEADLINE_CASES[k] for k in ("h8_768cat_512", "h4_384cat_256", "h1_up_128_64", "h1_64_64")},
             **CFG2_CASES, "ragged_17x33": (2, 17, 33, 48, 80, 0, 0), "odd_rows_50x18": (1, 50, 18, 32, 64, 16, 0)}


@pytest.mark.parametrize("case", sorted(W16_CASES))
def test_conv3x3_wino4_16_region_form_equals_the_one_workgroup_form(ctx, case, entry="nd_conv3x3_wino4_16_nhwc_f32"):
    """nd_conv3x3_wino4_16_nhwc_f32 (16 x 16-pixel regions, two co-resident workgroups per CU) against nn.Conv2d AND against
    nd_conv3x3_wino4_nhwc_f32 bit for bit wherever that kernel takes the shape -- output and GroupNorm partials, plain / concat / nearest-x2 /
    affine + SiLU sources, ragged images, images narrower than 32 pixels, partial K chunks and cout tiles; bitwise repeat."""
    import hiputil as hu
    B, H, W, cin, cout, c0, up = W16_CASES[case]
    cin = max(cin, 24)
    hs, ws = (H // 2, W // 2) if up else (H, W)
    bound = 1.0 / np.sqrt(9 * cin)
    x = U(case + ".x", (B, cin, hs, ws), -1.5, 1.5)
    w = U(case + ".w", (cout, cin, 3, 3), -bound, bound)
    b = U(case + ".b", (cout,), -bound, bound)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    ref = F.conv2d(xin, w, b, padding=1)
    wd, bd = hu.dev(w), hu.dev(b)
    wp = hu.full((ctx.lib.nd_pack_conv3x3_wino4_weight_floats(cin, cout),))
    L.call("nd_pack_conv3x3_wino4_weight", wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream)
    ctx.sync()
    run16 = lambda s, stats=True: _run_wino4(ctx, s, wp, bd, B, H, W, cin, cout, stats, entry=entry)
    run32 = lambda s, stats=True: _run_wino4(ctx, s, wp, bd, B, H, W, cin, cout, stats)
    s = hu.src(hu.nhwc(x[:, :c0]), hu.nhwc(x[:, c0:])) if c0 else hu.src(hu.nhwc(x), upsample=up)
    out, st, sc, slots = run16(s)
    assert rel_err(hu.nchw(out), ref) < 5e-5
    _check_gn(ctx, case, ref, st, sc, slots, B, cout, 5e-5)
    out2, st2, *_ = run16(s)
    assert torch.equal(out.cpu(), out2.cpu()) and torch.equal(st.cpu(), st2.cpu())
    o32, s32, c32, _ = run32(s)                                          # same arithmetic in the same order: identical bits
    assert torch.equal(out.cpu(), o32.cpu()) and torch.equal(st.cpu(), s32.cpu()) and torch.equal(sc.cpu(), c32.cpu())
    out_ns, *_ = run16(s, stats=False)
    assert torch.equal(out.cpu(), out_ns.cpu())
    if not up:
        M, A, D = U(case + ".M", (B, cin)), U(case + ".A", (B, cin), 0.5, 1.5), U(case + ".D", (B, cin))
        act = F.silu((x - M[:, :, None, None]) * A[:, :, None, None] + D[:, :, None, None])
        mad = hu.dev(torch.stack((M, A, D), 1))
        sa = hu.src(hu.nhwc(x[:, :c0]), hu.nhwc(x[:, c0:]), L.PRO_AFFINE_SILU, mad=mad) if c0 else hu.src(hu.nhwc(x), None, L.PRO_AFFINE_SILU, mad=mad)
        out, st, *_ = run16(sa)
        assert rel_err(hu.nchw(out), F.conv2d(act, w, b, padding=1)) < 5e-5
        o32, s32, *_ = run32(sa)
        assert torch.equal(out.cpu(), o32.cpu()) and torch.equal(st.cpu(), s32.cpu())
    for mode in (L.PRO_LEAKY, L.PRO_AFFINE_MAP_SILU):                    # these prologues stay on the one-workgroup form
        d = L.Conv3x3()
        d.src, d.weight, d.bias, d.out = hu.src(hu.nhwc(x), None, mode), wp.data_ptr(), bd.data_ptr(), out.data_ptr()
        d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, hs, ws, cin, cout, cout
        assert getattr(ctx.lib, entry)(C.byref(d), ctx.stream) != 0


@pytest.mark.parametrize("case", sorted(CFG2_CASES))
def test_conv3x3_wino4_16_split_k_for_sampling(ctx, case):
    """nd_conv3x3_wino4_16_splitk_nhwc_f32 with the split count of nd_conv3x3_wino4_16_splitk_plan (the sample's geometry only) on BASELINE config 2's
    narrow stages: output against nn.Conv2d and against the unsplit kernel, the GroupNorm partials the reduction leaves, the affine + SiLU prologue,
    bitwise repeat, and the same bits for a sample whatever batch it sits in."""
    import hiputil as hu
    B, H, W, cin, cout, c0, up = CFG2_CASES[case]
    pack, entry, plain_entry = "nd_pack_conv3x3_wino4_weight", "nd_conv3x3_wino4_16_splitk_nhwc_f32", "nd_conv3x3_wino4_16_nhwc_f32"
    B = min(B, 4)
    hs, ws = (H // 2, W // 2) if up else (H, W)
    bound = 1.0 / np.sqrt(9 * cin)
    x = U(case + ".x", (B, cin, hs, ws), -1.5, 1.5)
    w = U(case + ".w", (cout, cin, 3, 3), -bound, bound)
    b = U(case + ".b", (cout,), -bound, bound)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    ref = F.conv2d(xin, w, b, padding=1)
    wd, bd = hu.dev(w), hu.dev(b)
    wp = hu.full((ctx.lib.nd_pack_conv3x3_wino4_weight_floats(cin, cout),))
    L.call(pack, wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream)
    ctx.sync()
    splits = ctx.lib.nd_conv3x3_wino4_16_splitk_plan(H, W, cin, cout)
    assert splits in (2, 4)
    slots = ctx.lib.nd_conv3x3_wino4_stat_slots(H, W)

    def run(s, nb=B):
        out, st, sc = hu.full((nb, H, W, cout)), hu.full((nb, slots, cout, 2)), hu.full((slots,))
        ws_ = hu.full((ctx.lib.nd_conv3x3_wino4_splitk_workspace_floats(nb, H, W, cout, splits),))
        d = L.Conv3x3()
        d.src, d.weight, d.bias, d.out, d.stats, d.slot_count = s, wp.data_ptr(), bd.data_ptr(), out.data_ptr(), st.data_ptr(), sc.data_ptr()
        d.B, d.H, d.W, d.cin, d.cout, d.ldo = nb, H, W, cin, cout, cout
        L.call(entry, C.byref(d), ws_.data_ptr(), splits, ctx.stream)
        ctx.sync()
        return out, st, sc

    mk = lambda xx, **kw: (hu.src(hu.nhwc(xx[:, :c0]), hu.nhwc(xx[:, c0:]), **kw) if c0 else hu.src(hu.nhwc(xx), upsample=up, **kw))
    out, st, sc = run(mk(x))
    assert rel_err(hu.nchw(out), ref) < 5e-5
    _check_gn(ctx, case, ref, st, sc, slots, B, cout, 5e-5)
    out2, st2, _ = run(mk(x))
    assert torch.equal(out.cpu(), out2.cpu()) and torch.equal(st.cpu(), st2.cpu())
    o1, s1, _ = run(mk(x[1:2].contiguous()), nb=1)                       # sample 1 alone: the same bits as inside the batch
    assert torch.equal(out[1:2].cpu(), o1.cpu()) and torch.equal(st[1:2].cpu(), s1.cpu())
    plain, *_ = _run_wino4(ctx, mk(x), wp, bd, B, H, W, cin, cout, entry=plain_entry)
    assert rel_err(hu.nchw(out), hu.nchw(plain)) < 5e-5                  # another summation order over cin, nothing else (measured 2.3e-5 at 24 chunks)
    if not up:
        M, A, D = U(case + ".M", (B, cin)), U(case + ".A", (B, cin), 0.5, 1.5), U(case + ".D", (B, cin))
        act = F.silu((x - M[:, :, None, None]) * A[:, :, None, None] + D[:, :, None, None])
        out, *_ = run(mk(x, mode=L.PRO_AFFINE_SILU, mad=hu.dev(torch.stack((M, A, D), 1))))
        assert rel_err(hu.nchw(out), F.conv2d(act, w, b, padding=1)) < 5e-5


@pytest.mark.parametrize("dim,B", [(64, 16), (16, 2), (128, 8), (48, 3)])
def test_cond_step_single_launch_matches_torch(ctx, dim, B):
    """nd_cond_step_f32 == SinusoidalPosEmb -> Linear -> GELU -> Linear -> SiLU -> stacked ResnetBlock.mlp Linears
    (Diffusion_arch.py:100-107, 502-507, 149-152), and equals the four separate launches it replaces."""
    import math
    import hiputil as hu
    J = 2 * (3 * dim + 40)                                               # any row count; not a multiple of the grid stride
    t = torch.tensor([(997 * i + 3) % 1000 for i in range(B)], dtype=torch.long)
    half = dim // 2
    freqs = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1))).to(torch.float32)
    W1, b1 = U(f"cs.W1.{dim}", (4 * dim, dim), -0.2, 0.2), U(f"cs.b1.{dim}", (4 * dim,))
    W2, b2 = U(f"cs.W2.{dim}", (4 * dim, 4 * dim), -0.1, 0.1), U(f"cs.b2.{dim}", (4 * dim,))
    Wp, bp = U(f"cs.Wp.{dim}", (J, 4 * dim), -0.1, 0.1), U(f"cs.bp.{dim}", (J,))
    ang = t[:, None].float() * freqs[None]
    emb = torch.cat((ang.sin(), ang.cos()), -1)
    ref = F.linear(F.silu(F.linear(F.gelu(F.linear(emb, W1, b1)), W2, b2)), Wp, bp)
    td, fd = hu.dev(t), hu.dev(freqs)
    dW1, db1, dW2, db2, dWp, dbp = (hu.dev(v) for v in (W1, b1, W2, b2, Wp, bp))
    out = hu.full((B, J + 3))
    assert ctx.lib.nd_cond_step_lds_bytes(B, dim) == B * dim * 9 * 4
    L.call("nd_cond_step_f32", td.data_ptr(), fd.data_ptr(), dW1.data_ptr(), db1.data_ptr(), dW2.data_ptr(), db2.data_ptr(),
           dWp.data_ptr(), dbp.data_ptr(), out.data_ptr(), J + 3, B, dim, J, ctx.stream)
    ctx.sync()
    assert rel_err(out[:, :J].cpu(), ref) < TOL
    assert torch.isnan(out[:, J:]).all()                                 # nothing written beyond the J columns
    # the launches it replaces
    e, t1, st, o2 = hu.full((B, dim)), hu.full((B, 4 * dim)), hu.full((B, 4 * dim)), hu.full((B, J))
    L.call("nd_sinusoidal_time_emb_f32", td.data_ptr(), fd.data_ptr(), e.data_ptr(), B, half, ctx.stream)
    L.call("nd_linear_rows_f32", e.data_ptr(), dim, dW1.data_ptr(), db1.data_ptr(), t1.data_ptr(), 4 * dim, B, dim, 4 * dim, 0, L.ACT_GELU, ctx.stream)
    L.call("nd_linear_rows_f32", t1.data_ptr(), 4 * dim, dW2.data_ptr(), db2.data_ptr(), st.data_ptr(), 4 * dim, B, 4 * dim, 4 * dim, 0, L.ACT_SILU, ctx.stream)
    L.call("nd_linear_rows_f32", st.data_ptr(), 4 * dim, dWp.data_ptr(), dbp.data_ptr(), o2.data_ptr(), J, B, 4 * dim, J, 0, 0, ctx.stream)
    ctx.sync()
    assert rel_err(out[:, :J].cpu(), o2.cpu()) < 1e-5
    with pytest.raises(L.HipError, match="LDS"):
        L.call("nd_cond_step_f32", td.data_ptr(), fd.data_ptr(), dW1.data_ptr(), db1.data_ptr(), dW2.data_ptr(), db2.data_ptr(),
               dWp.data_ptr(), dbp.data_ptr(), out.data_ptr(), J + 3, 64, 128, J, ctx.stream)
    # the head looked up in a per-timestep table (built once per weight set by the same kernel code): the very same bits as computing it,
    # also for a batch with a timestep beyond the table (the whole batch then falls back to computing)
    rows = 1000
    table = hu.full((rows, 4 * dim))
    L.call("nd_cond_table_build_f32", fd.data_ptr(), dW1.data_ptr(), db1.data_ptr(), dW2.data_ptr(), db2.data_ptr(), table.data_ptr(), rows, dim, ctx.stream)
    ctx.sync()
    assert not torch.isnan(table).any()
    for tt in (t, torch.where(torch.arange(B) == B - 1, torch.tensor(1234), t)):
        tdev = hu.dev(tt)
        a_, b_ = hu.full((B, J)), hu.full((B, J))
        L.call("nd_cond_step_f32", tdev.data_ptr(), fd.data_ptr(), dW1.data_ptr(), db1.data_ptr(), dW2.data_ptr(), db2.data_ptr(),
               dWp.data_ptr(), dbp.data_ptr(), a_.data_ptr(), J, B, dim, J, ctx.stream)
        L.call("nd_cond_step_table_f32", tdev.data_ptr(), fd.data_ptr(), dW1.data_ptr(), db1.data_ptr(), dW2.data_ptr(), db2.data_ptr(),
               dWp.data_ptr(), dbp.data_ptr(), b_.data_ptr(), J, B, dim, J, table.data_ptr(), rows, ctx.stream)
        ctx.sync()
        assert torch.equal(a_.cpu(), b_.cpu())
    # ... and the projection tabulated as well (row t = what the launch computes for timestep t): a copy of B rows with the same bits; a timestep beyond the table: computed
    if J % 4 == 0:
        ptable, ts = hu.full((rows, J)), hu.dev(torch.arange(rows, dtype=torch.int64))
        for t0 in range(0, rows, 16):
            nb = min(16, rows - t0)
            L.call("nd_cond_step_table_f32", ts.data_ptr() + 8 * t0, fd.data_ptr(), dW1.data_ptr(), db1.data_ptr(), dW2.data_ptr(), db2.data_ptr(),
                   dWp.data_ptr(), dbp.data_ptr(), ptable.data_ptr() + 4 * t0 * J, J, nb, dim, J, table.data_ptr(), rows, ctx.stream)
        ctx.sync()
        assert not torch.isnan(ptable).any()
        for tt in (t, torch.where(torch.arange(B) == B - 1, torch.tensor(1234), t)):
            tdev = hu.dev(tt)
            a_, b_ = hu.full((B, J)), hu.full((B, J))
            L.call("nd_cond_step_f32", tdev.data_ptr(), fd.data_ptr(), dW1.data_ptr(), db1.data_ptr(), dW2.data_ptr(), db2.data_ptr(),
                   dWp.data_ptr(), dbp.data_ptr(), a_.data_ptr(), J, B, dim, J, ctx.stream)
            L.call("nd_cond_step_ptable_f32", tdev.data_ptr(), fd.data_ptr(), dW1.data_ptr(), db1.data_ptr(), dW2.data_ptr(), db2.data_ptr(),
                   dWp.data_ptr(), dbp.data_ptr(), b_.data_ptr(), J, B, dim, J, table.data_ptr(), rows, ptable.data_ptr(), ctx.stream)
            ctx.sync()
            assert torch.equal(a_.cpu(), b_.cpu())
