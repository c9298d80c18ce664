"""Training slice on the GPU: conv3x3 forward / data gradient / weight gradient on the HIP library, as a
torch.autograd.Function, against PyTorch's own fp32 convolution autograd on the same tensors."""
import copy
import ctypes as C

import pytest
import torch
import torch.nn.functional as F
from torch import nn

pytestmark = pytest.mark.gpu

from noisediff_amd import _lib as L, synth, train
from util import rel_err

DEV = torch.device("cuda", 0)
# (B, H, W, cin, cout): F(4x4) forward + dgrad; F(2x2) (narrow image); direct kernel (8x8); uneven channel blocks, ragged tiles
CASES = {"wino4": (2, 32, 64, 64, 64), "wino4_wide": (1, 32, 32, 128, 192), "wino2": (2, 24, 20, 32, 48), "direct": (3, 8, 8, 16, 24),
         "ragged": (2, 40, 36, 24, 72)}


def U(name, shape, lo=-1.0, hi=1.0):
    return synth.uniform(11, name, shape, lo, hi)


@pytest.mark.parametrize("case", sorted(CASES))
def test_conv3x3_autograd_matches_torch(case):
    B, H, W, cin, cout = CASES[case]
    x = U(case + ".x", (B, cin, H, W), -1.5, 1.5).to(DEV)
    w = (U(case + ".w", (cout, cin, 3, 3)) / (9 * cin) ** 0.5).to(DEV)
    b = U(case + ".b", (cout,)).to(DEV)
    gy = U(case + ".gy", (B, cout, H, W)).to(DEV)
    outs = []
    for fn in (lambda a, ww, bb: F.conv2d(a, ww, bb, padding=1), train.conv3x3):
        xa, wa, ba = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
        y = fn(xa, wa, ba)
        y.backward(gy)
        outs.append([t.detach().float().cpu() for t in (y, xa.grad, wa.grad, ba.grad)])
    for name, ref, got in zip(("y", "grad_x", "grad_w", "grad_b"), *outs):
        assert got.shape == ref.shape
        assert rel_err(got.numpy(), ref.numpy()) < 1e-4, (case, name)     # F(4x4) transforms: ~1e-5; MIOpen's own summation order differs too
    # no bias, input that needs no gradient
    xa, wa = x.clone(), w.clone().requires_grad_()
    train.conv3x3(xa, wa).backward(gy)
    assert rel_err(wa.grad.cpu().numpy(), outs[0][2].numpy()) < 1e-4


WGRAD_FORMS = {1: "nine taps", 2: "Winograd domain, four waves", 3: "Winograd domain, eight waves"}


@pytest.fixture
def wgrad_form():
    """Pins a form of nd_conv3x3_wgrad_nhwc_f32 for a test and restores the product's choice (by the shape) afterwards."""
    lib = L.load()
    was = lib.nd_conv3x3_wgrad_form(-1)
    yield lib.nd_conv3x3_wgrad_form
    lib.nd_conv3x3_wgrad_form(was)


@pytest.mark.parametrize("form", sorted(WGRAD_FORMS))
@pytest.mark.parametrize("shape", [(2, 48, 32, 64, 128), (2, 48, 24, 64, 128), (4, 64, 64, 32, 32), (1, 16, 16, 512, 256), (3, 40, 48, 96, 64), (2, 48, 32, 48, 64),
                                   (1, 4, 16, 32, 64), (2, 12, 16, 32, 32), (1, 6, 16, 32, 32), (3, 64, 256, 32, 64), (1, 16, 32, 96, 48), (2, 16, 16, 48, 48),
                                   (2, 8, 32, 80, 112)])
def test_wgrad_is_bitwise_repeatable_and_matches_a_float64_sum(shape, form, wgrad_form):
    """nd_conv3x3_wgrad_nhwc_f32 in each of its forms -- nine taps (any shape), the Winograd-domain F(4x4) kernel on four waves (H % 4 == 0, W % 16 == 0,
    channel counts % 16 == 0 -- d = 48's 48 / 96 / 80 / 112: half-empty last blocks --; down to a single tile group per workgroup, 1 / 2 / 3 groups: the three exits of its pipeline) and on eight waves
    (cout % 64 == 0 as well) -- against a float64 weight gradient; bitwise repeatable, bias gradient, argument checks.  A form that does not take the
    shape leaves it to the next one, as the product does; the product's own choice (by the shape) is one of them."""
    lib = L.load()
    wgrad_form(form)
    B, H, W, cin, cout = shape
    x = U("wg.x", (B, cin, H, W)).to(DEV).contiguous(memory_format=torch.channels_last)
    gy = U("wg.gy", (B, cout, H, W)).to(DEV).contiguous(memory_format=torch.channels_last)
    ws = torch.empty(int(lib.nd_conv3x3_wgrad_workspace_floats(B, H, W, cin, cout)), device=DEV)
    outs, bouts = [], []
    for it in range(3):
        dw = torch.full((cout, cin, 3, 3), float("nan"), device=DEV)
        db = torch.full((cout,), float("nan"), device=DEV)
        torch.cuda.synchronize()
        L.call("nd_conv3x3_wgrad_nhwc_f32", x.data_ptr(), cin, gy.data_ptr(), cout, dw.data_ptr(), db.data_ptr() if it else None, ws.data_ptr(),
               B, H, W, cin, cout, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        outs.append(dw.cpu())
        bouts.append(db.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])            # with and without the bias gradient: the same bits
    assert bool(bouts[0].isnan().all()) and torch.equal(bouts[1], bouts[2])           # NULL dbias: untouched
    ref = torch.nn.grad.conv2d_weight(x.double().cpu(), (cout, cin, 3, 3), gy.double().cpu(), padding=1)
    assert rel_err(outs[0].numpy(), ref.numpy()) < 2e-5
    assert rel_err(bouts[1].numpy(), gy.double().cpu().sum(dim=(0, 2, 3)).numpy()) < (2e-6 if form == 1 else 5e-6)      # (Winograd forms: the bias rides on a transformed operand)
    assert lib.nd_conv3x3_wgrad_workspace_floats(0, 8, 8, 8, 8) == -1
    with pytest.raises(L.HipError):
        L.call("nd_conv3x3_wgrad_nhwc_f32", x.data_ptr(), cin, gy.data_ptr(), cout, dw.data_ptr(), None, ws.data_ptr(), B, H, W, 6, cout, None)


def test_wgrad_forms_are_chosen_by_the_shape_alone():
    """The product's choice (form 0): the workspace asked for tells which form a shape gets -- per form S x cout x (36 cin + 1) or the nine-tap layout."""
    lib = L.load()
    assert lib.nd_conv3x3_wgrad_form(-1) == 0
    sizes = {}
    for shape in [(4, 128, 128, 192, 128), (4, 128, 128, 64, 64), (2, 48, 24, 64, 128)]:
        per = []
        for form in (0, 1, 2, 3):
            lib.nd_conv3x3_wgrad_form(form)
            per.append(int(lib.nd_conv3x3_wgrad_workspace_floats(*shape)))
        lib.nd_conv3x3_wgrad_form(0)
        sizes[shape] = per
    assert sizes[(4, 128, 128, 192, 128)][0] == sizes[(4, 128, 128, 192, 128)][3]            # large: eight waves
    assert sizes[(4, 128, 128, 64, 64)][0] == sizes[(4, 128, 128, 64, 64)][2]                # small: four waves
    assert len(set(sizes[(2, 48, 24, 64, 128)])) == 1                                       # W % 16 != 0: nine taps whatever is asked


@pytest.mark.parametrize("case", [
    # B, H, W, cin (first source), cin of a second concat source, cout, ldo, expected splits
    (4, 32, 32, 512, 0, 512, 512, 4), (2, 32, 32, 256, 0, 256, 256, 4), (4, 64, 64, 256, 0, 256, 260, 2), (1, 48, 40, 128, 128, 192, 192, 4),
    (1, 32, 32, 1024, 512, 64, 64, 8), (3, 20, 36, 128, 0, 100, 100, 2)])
def test_wino4_split_k_equals_the_plain_kernel_and_torch(case):
    """nd_conv3x3_wino4_splitk_nhwc_f32 (cin cut into 2 / 4 / 8 ranges, partial sums added in range order, bias in the reduction) against
    nd_conv3x3_wino4_nhwc_f32 on the same packed weights and against torch's fp32 convolution: ragged images, a concat source, a padded
    output stride, cout that is not a multiple of the 64-cout tile; bitwise repeatable; the plan is a function of the shape alone."""
    lib = L.load()
    B, H, W, c0, c1, cout, ldo, want = case
    cin = c0 + c1
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.nd_conv3x3_wino4_splitk_plan(B, H, W, cin, cout) == want
    x = U(f"sk.x.{case}", (B, H, W, cin)).to(DEV)
    xa, xb = (x[..., :c0].contiguous(), x[..., c0:].contiguous()) if c1 else (x, None)
    w = (U(f"sk.w.{case}", (cout, cin, 3, 3)) / (9 * cin) ** 0.5).to(DEV)
    b = U(f"sk.b.{case}", (cout,)).to(DEV)
    wp = torch.empty(int(lib.nd_pack_conv3x3_wino4_weight_floats(cin, cout)), device=DEV)
    L.call("nd_pack_conv3x3_wino4_weight", w.data_ptr(), wp.data_ptr(), cin, cout, st)
    d = L.Conv3x3()
    d.src.p0, d.src.c0, d.src.ld0, d.src.mode = xa.data_ptr(), c0, c0, L.PRO_NONE
    if c1:
        d.src.p1, d.src.c1, d.src.ld1 = xb.data_ptr(), c1, c1
    d.weight, d.bias = wp.data_ptr(), b.data_ptr()
    d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, ldo
    outs = []
    for split in (0, want, want):
        out = torch.full((B, H, W, ldo), float("nan"), device=DEV)
        d.out = out.data_ptr()
        if split:
            ws = torch.full((int(lib.nd_conv3x3_wino4_splitk_workspace_floats(B, H, W, cout, split)),), float("nan"), device=DEV)
            L.call("nd_conv3x3_wino4_splitk_nhwc_f32", C.byref(d), ws.data_ptr(), split, st)
        else:
            L.call("nd_conv3x3_wino4_nhwc_f32", C.byref(d), st)
        torch.cuda.synchronize()
        assert bool(out[..., cout:].isnan().all())                                   # the padding of the output stride is not touched
        outs.append(out[..., :cout].cpu())
    assert torch.equal(outs[1], outs[2])
    ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), b.double().cpu(), padding=1).permute(0, 2, 3, 1).float()
    assert rel_err(outs[0].numpy(), ref.numpy()) < 5e-5 and rel_err(outs[1].numpy(), ref.numpy()) < 5e-5
    assert rel_err(outs[1].numpy(), outs[0].numpy()) < 5e-5                          # the same products, another summation order over cin
    # rejected: statistics epilogue, prologues, split counts that do not divide the K chunks
    with pytest.raises(L.HipError):
        L.call("nd_conv3x3_wino4_splitk_nhwc_f32", C.byref(d), ws.data_ptr(), 3, st)
    d.src.mode = L.PRO_LEAKY
    with pytest.raises(L.HipError):
        L.call("nd_conv3x3_wino4_splitk_nhwc_f32", C.byref(d), ws.data_ptr(), want, st)


def test_wino4_split_k_with_the_affine_silu_prologue():
    """The split-K instance of the GroupNorm-affine + SiLU prologue (block2's convolution in the engine's low-latency mode) against the plain kernel
    and torch: conv(silu((x - M) A + D)) with per-(sample, channel) M, A, D, statistics on."""
    lib = L.load()
    B, H, W, cin, cout = 1, 32, 32, 512, 512
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    splits = lib.nd_conv3x3_wino4_splitk_plan(B, H, W, cin, cout)
    x = U("ska.x", (B, H, W, cin)).to(DEV)
    w = (U("ska.w", (cout, cin, 3, 3)) / (9 * cin) ** 0.5).to(DEV)
    b = U("ska.b", (cout,)).to(DEV)
    mad = torch.stack((U("ska.m", (B, cin)) * 0.2, U("ska.a", (B, cin), 0.5, 1.5), U("ska.d", (B, cin)) * 0.3), dim=1).contiguous().to(DEV)   # [B][3][cin]
    wp = torch.empty(int(lib.nd_pack_conv3x3_wino4_weight_floats(cin, cout)), device=DEV)
    L.call("nd_pack_conv3x3_wino4_weight", w.data_ptr(), wp.data_ptr(), cin, cout, st)
    slots = lib.nd_conv3x3_wino4_stat_slots(H, W)
    outs = []
    for split in (0, splits):
        out = torch.full((B, H, W, cout), float("nan"), device=DEV)
        stt, sc = torch.empty((B, slots, cout, 2), device=DEV), torch.empty((slots,), device=DEV)
        d = L.Conv3x3()
        d.src.p0, d.src.c0, d.src.ld0, d.src.mode, d.src.mad = x.data_ptr(), cin, cin, L.PRO_AFFINE_SILU, mad.data_ptr()
        d.weight, d.bias, d.out, d.stats, d.slot_count = wp.data_ptr(), b.data_ptr(), out.data_ptr(), stt.data_ptr(), sc.data_ptr()
        d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
        if split:
            ws = torch.empty((int(lib.nd_conv3x3_wino4_splitk_workspace_floats(B, H, W, cout, split)),), device=DEV)
            L.call("nd_conv3x3_wino4_splitk_nhwc_f32", C.byref(d), ws.data_ptr(), split, st)
        else:
            L.call("nd_conv3x3_wino4_nhwc_f32", C.byref(d), st)
        torch.cuda.synchronize()
        outs.append((out.cpu(), stt.cpu()))
    assert splits == 8 and rel_err(outs[1][0].numpy(), outs[0][0].numpy()) < 5e-5
    assert rel_err(outs[1][1][..., 0].numpy(), outs[0][1][..., 0].numpy()) < 1e-4
    xin = F.silu((x.double().cpu() - mad[:, 0].double().cpu()[:, None, None, :]) * mad[:, 1].double().cpu()[:, None, None, :] + mad[:, 2].double().cpu()[:, None, None, :])
    ref = F.conv2d(xin.permute(0, 3, 1, 2), w.double().cpu(), b.double().cpu(), padding=1).permute(0, 2, 3, 1).float()
    assert rel_err(outs[1][0].numpy(), ref.numpy()) < 5e-5


@pytest.mark.parametrize("case", [(1, 32, 32, 512, 512, 8), (2, 48, 40, 256, 192, 8), (1, 64, 64, 256, 100, 4)])
def test_wino4_split_k_statistics_match_the_plain_epilogue(case):
    """The split-K form with a statistics epilogue (the reduction kernel leaves the slots of the summed output) against nd_conv3x3_wino4_nhwc_f32's own
    epilogue: same slot layout and counts, sums and centred second moments equal within rounding, and the GroupNorm coefficients pooled from them
    (nd_groupnorm_finalize_f32) equal to 1e-5 -- ragged tiles, cout that is not a multiple of 64."""
    lib = L.load()
    B, H, W, cin, cout, groups = case
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    splits = lib.nd_conv3x3_wino4_splitk_plan(B, H, W, cin, cout)
    assert splits > 1
    x = U(f"sks.x.{case}", (B, H, W, cin)).to(DEV)
    w = (U(f"sks.w.{case}", (cout, cin, 3, 3)) / (9 * cin) ** 0.5).to(DEV)
    b = U(f"sks.b.{case}", (cout,)).to(DEV)
    gam, bet = U(f"sks.g.{case}", (cout,), 0.5, 1.5).to(DEV), U(f"sks.be.{case}", (cout,)).to(DEV)
    wp = torch.empty(int(lib.nd_pack_conv3x3_wino4_weight_floats(cin, cout)), device=DEV)
    L.call("nd_pack_conv3x3_wino4_weight", w.data_ptr(), wp.data_ptr(), cin, cout, st)
    slots = lib.nd_conv3x3_wino4_stat_slots(H, W)
    res = []
    for split in (0, splits):
        out = torch.full((B, H, W, cout), float("nan"), device=DEV)
        stt = torch.full((B, slots, cout, 2), float("nan"), device=DEV)
        sc = torch.full((slots,), float("nan"), device=DEV)
        mad = torch.full((B, 3, cout), float("nan"), device=DEV)
        d = L.Conv3x3()
        d.src.p0, d.src.c0, d.src.ld0, d.src.mode = x.data_ptr(), cin, cin, L.PRO_NONE
        d.weight, d.bias, d.out, d.stats, d.slot_count = wp.data_ptr(), b.data_ptr(), out.data_ptr(), stt.data_ptr(), sc.data_ptr()
        d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
        if split:
            ws = torch.full((int(lib.nd_conv3x3_wino4_splitk_workspace_floats(B, H, W, cout, split)),), float("nan"), device=DEV)
            L.call("nd_conv3x3_wino4_splitk_nhwc_f32", C.byref(d), ws.data_ptr(), split, st)
        else:
            L.call("nd_conv3x3_wino4_nhwc_f32", C.byref(d), st)
        L.call("nd_groupnorm_finalize_f32", stt.data_ptr(), sc.data_ptr(), slots, gam.data_ptr(), bet.data_ptr(), None, 0, mad.data_ptr(), B, cout, groups, 1e-5, st)
        torch.cuda.synchronize()
        res.append((out.cpu(), stt.cpu(), sc.cpu(), mad.cpu()))
    (o0, s0, c0, m0), (o1, s1, c1, m1) = res
    assert not bool(o1.isnan().any()) and not bool(s1.isnan().any()) and torch.equal(c0, c1)
    assert rel_err(o1.numpy(), o0.numpy()) < 5e-5
    assert rel_err(s1[..., 0].numpy(), s0[..., 0].numpy()) < 1e-4 and rel_err(s1[..., 1].numpy(), s0[..., 1].numpy()) < 1e-3
    assert rel_err(m1.numpy(), m0.numpy()) < 1e-5
    xr = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), b.double().cpu(), padding=1)
    ref = F.group_norm(xr, groups, gam.double().cpu(), bet.double().cpu(), 1e-5).permute(0, 2, 3, 1)
    got = (o1.double() - m1[:, 0].double()[:, None, None, :]) * m1[:, 1].double()[:, None, None, :] + m1[:, 2].double()[:, None, None, :]
    assert rel_err(got.float().numpy(), ref.float().numpy()) < 1e-4


def test_modulate_silu_and_the_fused_shortcut_match_torch():
    """train.modulate_silu (ResnetBlock2's per-pixel modulation + SiLU, forward and backward) and group_norm_silu(..., res=) against the same
    expressions in torch (float64 reference)."""
    B, C_, H, W = 2, 64, 24, 20
    n = U("ms.n", (B, C_, H, W)).to(DEV).contiguous(memory_format=torch.channels_last)
    ss = U("ms.ss", (B, 2 * C_, H, W)).to(DEV).contiguous(memory_format=torch.channels_last)
    gy = U("ms.g", (B, C_, H, W)).to(DEV)
    na, sa = n.clone().requires_grad_(), ss.clone().requires_grad_()
    y = train.modulate_silu(na, sa)
    y.backward(gy)
    nd, sd = n.double().requires_grad_(), ss.double().requires_grad_()
    sc, sh = sd.chunk(2, dim=1)
    yr = F.silu(nd * (sc + 1) + sh)
    yr.backward(gy.double())
    for name, got, ref in (("y", y, yr), ("dn", na.grad, nd.grad), ("dss", sa.grad, sd.grad)):
        assert rel_err(got.detach().cpu().numpy(), ref.detach().float().cpu().numpy()) < 2e-6, name
    with pytest.raises(ValueError):
        train.modulate_silu(n, ss[:, :C_])
    # the ResnetBlock's shortcut inside the fused norm + SiLU operator
    x = U("ms.x", (B, C_, H, W)).to(DEV)
    res = U("ms.res", (B, C_, H, W)).to(DEV)
    gam, bet = U("ms.gam", (C_,), 0.5, 1.5).to(DEV), U("ms.bet", (C_,)).to(DEV)
    tss = U("ms.tss", (B, 2 * C_)).to(DEV)
    xa, ra, ga, ba, ta = (t.clone().requires_grad_() for t in (x, res, gam, bet, tss))
    y = train.group_norm_silu(xa, 8, ga, ba, ta, 1e-5, res=ra)
    y.backward(gy)
    xd, rd, gd_, bd, td = (t.double().requires_grad_() for t in (x, res, gam, bet, tss))
    sc, sh = td[:, :, None, None].chunk(2, dim=1)
    yr = F.silu(F.group_norm(xd, 8, gd_, bd, 1e-5) * (sc + 1) + sh) + rd
    yr.backward(gy.double())
    for name, got, ref in (("y", y, yr), ("dx", xa.grad, xd.grad), ("dres", ra.grad, rd.grad), ("dgamma", ga.grad, gd_.grad), ("dbeta", ba.grad, bd.grad),
                           ("dss", ta.grad, td.grad)):
        assert rel_err(got.detach().cpu().numpy(), ref.detach().float().cpu().numpy()) < 2e-5, name


@pytest.mark.parametrize("case", [(2, 64, 64, 64, 64, 8), (4, 32, 32, 512, 512, 8), (2, 16, 16, 32, 64, 2), (1, 8, 8, 16, 16, 8), (2, 40, 36, 24, 40, 8)])
def test_block_with_the_convs_statistics_epilogue_matches_torch(case):
    """conv3x3_with_stats + group_norm_silu(conv_stats=): Block (conv -> GroupNorm -> modulation -> SiLU (+ shortcut)) with the norm's moments taken from
    the convolution kernel's statistics epilogue (all three forward kernels and the split-K form's reduction) against torch in
    float64, forward and every gradient."""
    B, H, W, cin, cout, groups = case
    x = U(f"cs.x.{case}", (B, cin, H, W)).to(DEV)
    w = (U(f"cs.w.{case}", (cout, cin, 3, 3)) / (9 * cin) ** 0.5).to(DEV)
    b = U(f"cs.b.{case}", (cout,)).to(DEV)
    gam, bet = U(f"cs.g.{case}", (cout,), 0.5, 1.5).to(DEV), U(f"cs.be.{case}", (cout,)).to(DEV)
    ss = U(f"cs.ss.{case}", (B, 2 * cout)).to(DEV)
    res = U(f"cs.r.{case}", (B, cout, H, W)).to(DEV)
    gy = U(f"cs.gy.{case}", (B, cout, H, W)).to(DEV)
    ts = [t.clone().requires_grad_() for t in (x, w, b, gam, bet, ss, res)]
    y, cs = train.conv3x3_with_stats(ts[0], ts[1], ts[2])
    assert cs is not None and cs[0].shape[0] == B and cs[0].shape[2:] == (cout, 2)      # (split-K layers included: their reduction kernel leaves the slots)
    out = train.group_norm_silu(y, groups, ts[3], ts[4], ts[5], 1e-5, res=ts[6], conv_stats=cs)
    out.backward(gy)
    td = [t.double().requires_grad_() for t in (x, w, b, gam, bet, ss, res)]
    sc, sh = td[5][:, :, None, None].chunk(2, dim=1)
    ref = F.silu(F.group_norm(F.conv2d(td[0], td[1], td[2], padding=1), groups, td[3], td[4], 1e-5) * (sc + 1) + sh) + td[6]
    ref.backward(gy.double())
    assert rel_err(out.detach().cpu().numpy(), ref.detach().float().cpu().numpy()) < 1e-4
    for name, got, r in zip(("dx", "dw", "db", "dgamma", "dbeta", "dss", "dres"), ts, td):
        assert rel_err(got.grad.cpu().numpy(), r.grad.float().cpu().numpy()) < 2e-4, (case, name)


def test_broadcast_add_token_sum_matches_a_float64_sum():
    """train.broadcast_add: tokens + a per-sample vector, the vector's gradient = nd_token_sum_f32 over the tokens (fixed order, repeatable)."""
    for (B, N, C_) in ((4, 65536, 64), (2, 1000, 128), (3, 50, 512)):
        t = U(f"ba.t.{N}", (B, N, C_)).to(DEV)
        v = U(f"ba.v.{N}", (B, 1, C_)).to(DEV)
        gy = U(f"ba.g.{N}", (B, N, C_)).to(DEV)
        outs = []
        for _ in range(2):
            ta, va = t.clone().requires_grad_(), v.clone().requires_grad_()
            y = train.broadcast_add(ta, va)
            y.backward(gy)
            outs.append((y.detach().cpu(), ta.grad.cpu(), va.grad.cpu()))
        assert all(torch.equal(a, b) for a, b in zip(*outs))
        assert torch.equal(outs[0][0], (t + v).cpu()) and torch.equal(outs[0][1], gy.cpu())
        assert rel_err(outs[0][2].numpy(), gy.double().sum(dim=1, keepdim=True).float().cpu().numpy()) < 2e-6


def test_linear_on_the_pointwise_kernels_matches_the_library_gemm():
    """ND_TRAIN_PW=1: output and data gradient of a token Linear on nd_pointwise_gemm_nhwc_f32 (data-gradient weight packed in place by
    nd_pack_pointwise_weight_t) against torch, ragged token counts and channel counts that are not multiples of the tiles included."""
    old = train._PW_GEMM
    train._PW_GEMM = True
    try:
        for (shape, cin, cout) in (((2, 1000, 64), 64, 128), ((4100, 72), 72, 40), ((3, 16), 16, 64), ((1, 8192, 256), 256, 256)):
            x = U(f"pwl.x.{cin}", shape).to(DEV)
            w = (U(f"pwl.w.{cin}", (cout, cin)) / cin ** 0.5).to(DEV)
            b = U(f"pwl.b.{cin}", (cout,)).to(DEV)
            gy = U(f"pwl.g.{cin}", shape[:-1] + (cout,)).to(DEV)
            outs = []
            for fn in (F.linear, train.linear):
                xa, wa, ba = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
                y = fn(xa, wa, ba)
                y.backward(gy)
                outs.append([t.detach().cpu() for t in (y, xa.grad, wa.grad, ba.grad)])
            for name, ref, got in zip(("y", "grad_x", "grad_w", "grad_b"), *outs):
                assert got.shape == ref.shape and rel_err(got.numpy(), ref.numpy()) < 2e-5, (shape, name)
    finally:
        train._PW_GEMM = old


@pytest.mark.parametrize("kind", ["", "_wino", "_wino4"])
def test_dgrad_packing_equals_packing_the_flipped_transposed_weight(kind):
    """nd_pack_conv3x3*_weight_dgrad(w) == nd_pack_conv3x3*_weight(w.flip(2, 3).transpose(0, 1)) bit for bit, ragged channel counts included."""
    lib = L.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for cout_f, cin_f in ((64, 128), (40, 72), (8, 24)):                              # forward layer: cin_f -> cout_f; its data gradient: cout_f -> cin_f
        w = U(f"dg.{kind}.{cout_f}", (cout_f, cin_f, 3, 3)).to(DEV)
        wt = w.flip(2, 3).transpose(0, 1).contiguous()
        n = int(getattr(lib, f"nd_pack_conv3x3{kind}_weight_floats")(cout_f, cin_f))
        a, b = torch.full((n,), float("nan"), device=DEV), torch.full((n,), float("nan"), device=DEV)
        L.call(f"nd_pack_conv3x3{kind}_weight", wt.data_ptr(), a.data_ptr(), cout_f, cin_f, st)
        L.call(f"nd_pack_conv3x3{kind}_weight_dgrad", w.data_ptr(), b.data_ptr(), cout_f, cin_f, st)
        torch.cuda.synchronize()
        assert torch.equal(a.cpu(), b.cpu()) and not bool(a.isnan().any()), (kind, cout_f, cin_f)


class _Block(nn.Module):
    """Conv3x3 -> GroupNorm -> SiLU -> Conv3x3 + 1x1 shortcut: the shape of the reference's ResnetBlock (Diffusion_arch.py:146-170)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.c1, self.n1 = nn.Conv2d(cin, cout, 3, padding=1), nn.GroupNorm(8, cout)
        self.c2, self.n2 = nn.Conv2d(cout, cout, 3, padding=1), nn.GroupNorm(8, cout)
        self.res = nn.Conv2d(cin, cout, 1)

    def forward(self, x):
        h = F.silu(self.n1(self.c1(x)))
        return F.silu(self.n2(self.c2(h))) + self.res(x)


def test_accelerate_keeps_losses_and_gradients_of_a_resnet_block():
    torch.manual_seed(0)
    ref = nn.Sequential(_Block(32, 64), _Block(64, 64), nn.Conv2d(64, 4, 1)).to(DEV)
    hip = copy.deepcopy(ref)
    assert train.accelerate(hip) == 4 and train.accelerate(hip) == 0           # four 3x3 convs taken, idempotent
    assert list(hip.state_dict()) == list(ref.state_dict())
    ema = copy.deepcopy(hip)                                                     # the trainer's EMA deep-copies the net
    assert ema[0].c1.forward.__self__ is ema[0].c1
    x = U("blk.x", (2, 32, 32, 64)).to(DEV)
    target = U("blk.t", (2, 4, 32, 64)).to(DEV)
    losses = []
    for net in (ref, hip):
        loss = F.mse_loss(net(x), target)
        loss.backward()
        losses.append(float(loss))
    assert losses[1] == pytest.approx(losses[0], rel=2e-5)
    for (k, p), q in zip(ref.named_parameters(), hip.parameters()):
        assert rel_err(q.grad.cpu().numpy(), p.grad.cpu().numpy()) < 2e-4, k
    with pytest.raises(L.HipError, match="no CPU path"):
        train.conv3x3(torch.zeros(1, 8, 8, 8), torch.zeros(8, 8, 3, 3))


GN_CASES = {"full_res": (2, 64, 8, 96, 64), "wide": (3, 256, 8, 16, 16), "one_group": (2, 32, 1, 20, 12), "narrow_rows": (2, 8, 2, 9, 7),
            "c1024": (1, 1024, 32, 8, 8)}


@pytest.mark.parametrize("case", sorted(GN_CASES))
def test_group_norm_autograd_matches_torch(case):
    """nd_groupnorm_train_forward / _backward == F.group_norm and its autograd (output, dx, dgamma, dbeta) on channels_last and on
    NCHW-contiguous inputs, inputs with a large mean (the pivot keeps the variance well conditioned), bitwise repeatable."""
    B, C_, G, H, W = GN_CASES[case]
    x = (U(case + ".x", (B, C_, H, W), -1.5, 1.5) + 3.0 * U(case + ".m", (B, C_, 1, 1))).to(DEV)
    gamma, beta = U(case + ".g", (C_,), 0.5, 1.5).to(DEV), U(case + ".b", (C_,)).to(DEV)
    gy = U(case + ".gy", (B, C_, H, W)).to(DEV)
    outs = []
    for fn, fmt in ((lambda a, w, b: F.group_norm(a, G, w, b, 1e-5), torch.contiguous_format),
                    (lambda a, w, b: train.group_norm(a, G, w, b, 1e-5), torch.channels_last),
                    (lambda a, w, b: train.group_norm(a, G, w, b, 1e-5), torch.contiguous_format)):
        xa = x.clone().contiguous(memory_format=fmt).requires_grad_()
        wa, ba = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
        y = fn(xa, wa, ba)
        y.backward(gy)
        outs.append([t.detach().float().cpu().contiguous() for t in (y, xa.grad, wa.grad, ba.grad)])
    ref64 = []
    xa, wa, ba = x.double().cpu().requires_grad_(), gamma.double().cpu().requires_grad_(), beta.double().cpu().requires_grad_()
    y = F.group_norm(xa, G, wa, ba, 1e-5)
    y.backward(gy.double().cpu())
    ref64 = [t.detach() for t in (y, xa.grad, wa.grad, ba.grad)]
    for i, name in enumerate(("y", "dx", "dgamma", "dbeta")):
        for got in (outs[1][i], outs[2][i]):
            assert got.shape == ref64[i].shape
            assert rel_err(got.numpy(), ref64[i].numpy()) < 2e-5, (case, name)
        assert torch.equal(outs[1][i], outs[2][i]), (case, name)                  # the memory format of the input does not change the bits
    with pytest.raises(ValueError):
        train.group_norm(x[:, :6], 2, gamma[:6], beta[:6])                          # C % 4 != 0: not taken


def test_accelerate_takes_the_norms_too():
    torch.manual_seed(1)
    ref = nn.Sequential(_Block(32, 64), _Block(64, 64), nn.Conv2d(64, 4, 1)).to(DEV).to(memory_format=torch.channels_last)
    hip, convs_only = copy.deepcopy(ref), copy.deepcopy(ref)
    train.accelerate(hip)
    train.accelerate(convs_only, norms=False)
    assert sum(getattr(m.forward, "__func__", None) is train._hip_norm_forward for m in hip.modules()) == 4
    assert sum(getattr(m.forward, "__func__", None) is train._hip_norm_forward for m in convs_only.modules()) == 0
    x = U("nrm.x", (2, 32, 48, 32)).to(DEV)
    target = U("nrm.t", (2, 4, 48, 32)).to(DEV)
    grads = []
    for net in (ref, hip):
        F.mse_loss(net(x), target).backward()
        grads.append({k: p.grad.detach().cpu() for k, p in net.named_parameters()})
    for k in grads[0]:
        assert rel_err(grads[1][k].numpy(), grads[0][k].numpy()) < 2e-4, k


LIN_CASES = {"tokens": (3 * 1000, 64, 128), "ragged": (777, 36, 20), "wide": (4096, 512, 1024), "one_token_per_sample": (4, 16, 64),
             "many": (262144, 64, 64)}


@pytest.mark.parametrize("case", sorted(LIN_CASES))
def test_linear_weight_and_bias_gradient_match_a_float64_sum(case):
    """nd_linear_wgrad_f32 through train.linear: dW and db against float64, dX and y against F.linear; bitwise repeatable."""
    N, cin, cout = LIN_CASES[case]
    x = U(case + ".x", (N, cin), -1.5, 1.5).to(DEV)
    w = (U(case + ".w", (cout, cin)) / cin ** 0.5).to(DEV)
    b = U(case + ".b", (cout,)).to(DEV)
    gy = U(case + ".gy", (N, cout)).to(DEV)
    outs = []
    for _ in range(2):
        xa, wa, ba = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
        y = train.linear(xa, wa, ba)
        y.backward(gy)
        outs.append([t.detach().cpu() for t in (y, xa.grad, wa.grad, ba.grad)])
    assert all(torch.equal(p, q) for p, q in zip(*outs))
    x64, g64 = x.double().cpu(), gy.double().cpu()
    assert rel_err(outs[0][2].numpy(), (g64.T @ x64).numpy()) < 2e-5 * max(1.0, (N / 4096) ** 0.5)
    assert rel_err(outs[0][3].numpy(), g64.sum(0).numpy()) < 2e-5 * max(1.0, (N / 4096) ** 0.5)
    assert rel_err(outs[0][1].numpy(), (g64 @ w.double().cpu()).numpy()) < 2e-5
    assert rel_err(outs[0][0].numpy(), F.linear(x64, w.double().cpu(), b.double().cpu()).numpy()) < 2e-5
    wa = w.clone().requires_grad_()                                         # no bias
    train.linear(x, wa).backward(gy)
    assert torch.equal(wa.grad.cpu(), outs[0][2])


def test_linear_and_conv1x1_with_the_residual_in_the_gemm_epilogue():
    """train.linear(x, w, b, res) / train.conv1x1(x, w, b, res=): y = x W^T + b + res with the addition in the GEMM's epilogue (the AttnBlock's feed-forward and
    proj_out residuals, Diffusion_arch.py:440-443): output and every gradient -- the residual's is the output's -- against F.linear / F.conv2d + res; the residual
    must have the output's shape."""
    x, w, b = U("lres.x", (2, 1536, 128)).to(DEV), (U("lres.w", (64, 128)) / 128 ** 0.5).to(DEV), U("lres.b", (64,)).to(DEV)
    r, gy = U("lres.r", (2, 1536, 64)).to(DEV), U("lres.gy", (2, 1536, 64)).to(DEV)
    outs = []
    for fused in (False, True):
        xa, wa, ba, ra = (t.clone().requires_grad_() for t in (x, w, b, r))
        y = train.linear(xa, wa, ba, ra) if fused else F.linear(xa, wa, ba) + ra
        y.backward(gy)
        outs.append([t.detach().cpu() for t in (y, xa.grad, wa.grad, ba.grad, ra.grad)])
    for got, ref, name in zip(outs[1], outs[0], ("y", "dx", "dw", "db", "dres")):
        assert rel_err(got.numpy(), ref.numpy()) < 2e-5, name
    assert torch.equal(outs[1][4], gy.cpu())
    with pytest.raises(ValueError, match="residual"):
        train.linear(x, w, b, r[:, :8])
    xc = U("cres.x", (2, 128, 32, 48)).to(DEV).contiguous(memory_format=torch.channels_last)
    wc = (U("cres.w", (64, 128, 1, 1)) / 128 ** 0.5).to(DEV)
    rc, gc = (U(n, (2, 64, 32, 48)).to(DEV).contiguous(memory_format=torch.channels_last) for n in ("cres.r", "cres.g"))
    outs = []
    for fused in (False, True):
        xa, wa, ba, ra = (t.clone().requires_grad_() for t in (xc, wc, b, rc))
        y = train.conv1x1(xa, wa, ba, res=ra) if fused else F.conv2d(xa, wa, ba) + ra
        y.backward(gc)
        outs.append([t.detach().cpu() for t in (y, xa.grad, wa.grad, ba.grad, ra.grad)])
    for got, ref, name in zip(outs[1], outs[0], ("y", "dx", "dw", "db", "dres")):
        assert rel_err(got.numpy(), ref.numpy()) < 2e-5, name


def test_conv1x1_and_accelerated_linears():
    torch.manual_seed(2)
    ref = nn.Sequential(nn.Conv2d(32, 64, 1), nn.SiLU(), nn.Conv2d(64, 64, 3, padding=1), nn.GroupNorm(8, 64), nn.Conv2d(64, 8, 1)).to(DEV)
    hip = copy.deepcopy(ref)
    train.accelerate(hip)
    assert sum(getattr(m.forward, "__func__", None) is train._hip_linear_forward for m in hip.modules()) == 2
    x = U("c11.x", (2, 32, 24, 40)).to(DEV)
    target = U("c11.t", (2, 8, 24, 40)).to(DEV)
    grads = []
    for net in (ref, hip):
        xa = x.clone().requires_grad_()
        F.mse_loss(net(xa), target).backward()
        grads.append({**{k: p.grad.detach().cpu() for k, p in net.named_parameters()}, "x": xa.grad.detach().cpu()})
    for k in grads[0]:
        assert rel_err(grads[1][k].numpy(), grads[0][k].numpy()) < 2e-4, k
    tok = nn.Linear(48, 96).to(DEV)
    tok_hip = copy.deepcopy(tok)
    train.accelerate(tok_hip)
    t = U("tok.x", (3, 50, 48)).to(DEV)
    for net in (tok, tok_hip):
        net(t).square().mean().backward()
    assert rel_err(tok_hip.weight.grad.cpu().numpy(), tok.weight.grad.cpu().numpy()) < 2e-5
    assert rel_err(tok_hip.bias.grad.cpu().numpy(), tok.bias.grad.cpu().numpy()) < 2e-5


@pytest.mark.parametrize("shape", [(2, 32, 48, 64), (1, 40, 24, 48), (3, 16, 16, 16), (2, 32, 64, 64), (1, 40, 32, 48), (3, 16, 96, 128), (2, 64, 128, 96), (1, 8, 32, 32),
                                   (5, 128, 128, 64)])
def test_stem_conv7x7_autograd_matches_pytorch(shape):
    """train.conv7x7_c4 (init_conv / cond_init_conv: forward on nd_conv7x7_c4_f32, weight + bias gradient on nd_conv7x7_c4_wgrad_f32 -- H % 4 == 0, W % 32 == 0,
    cout in {32, 48, 64, 96, 128}; one and several tiles per workgroup -- or, for the other shapes, through nd_linear_wgrad_f32 over the unfolded image)
    == F.conv2d(padding=3) and its autograd; the image's own gradient (not needed by the reference's training loop) comes from PyTorch's transposed convolution."""
    B, H, W, cout = shape
    x = U("stem.x", (B, 4, H, W), -1.5, 1.5).to(DEV)
    w, b = U("stem.w", (cout, 4, 7, 7), -0.1, 0.1).to(DEV), U("stem.b", (cout,)).to(DEV)
    gy = U("stem.gy", (B, cout, H, W)).to(DEV)
    outs = []
    for hip in (False, True, True):
        xa, wa, ba = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
        y = train.conv7x7_c4(xa, wa, ba) if hip else F.conv2d(xa, wa, ba, padding=3)
        y.backward(gy)
        outs.append([t.detach().cpu() for t in (y, xa.grad, wa.grad, ba.grad)])
    for got, ref, name in zip(outs[1], outs[0], ("y", "dx", "dw", "db")):
        assert rel_err(got.numpy(), ref.numpy()) < 2e-5 * max(1.0, (B * H * W / 4096) ** 0.5), name
    for p, q, name in zip(outs[1], outs[2], ("y", "dx", "dw", "db")):
        if name != "dx":                                                         # (dx is PyTorch's transposed convolution: MIOpen may pick an atomic algorithm)
            assert torch.equal(p, q), name                                        # bitwise repeatable
    with pytest.raises(ValueError):
        train.conv7x7_c4(x[:, :3], w[:, :3], b)


LN_CASES = {"c64": (3001, 64), "c128": (1000, 128), "c256": (515, 256), "c512": (300, 512), "c1024": (70, 1024), "many": (262144, 64),
            "c48": (3001, 48), "c96": (1000, 96), "c192": (777, 192), "c384": (301, 384), "c20": (513, 20), "c1000": (65, 1000)}      # (d = 48: 48 / 96 / 192 / 384; ragged widths)


@pytest.mark.parametrize("case", sorted(LN_CASES))
def test_layer_norm_autograd_matches_a_float64_reference(case):
    """nd_layernorm_train_forward / _backward == F.layer_norm and its autograd in float64 (y, dx, dgamma, dbeta); row counts that do not fill
    the last wave / workgroup, rows with a large mean, bitwise repeatable."""
    N, C_ = LN_CASES[case]
    x = (U(case + ".x", (N, C_), -1.5, 1.5) + 3.0 * U(case + ".m", (N, 1))).to(DEV)
    gamma, beta = U(case + ".g", (C_,), 0.5, 1.5).to(DEV), U(case + ".b", (C_,)).to(DEV)
    gy = U(case + ".gy", (N, C_)).to(DEV)
    outs = []
    for _ in range(2):
        xa, wa, ba = x.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
        y = train.layer_norm(xa.view(1, N, C_), wa, ba, 1e-5)
        assert y.shape == (1, N, C_)
        y.backward(gy.view(1, N, C_))
        outs.append([t.detach().cpu() for t in (y.view(N, C_), xa.grad, wa.grad, ba.grad)])
    assert all(torch.equal(p, q) for p, q in zip(*outs))
    xa, wa, ba = x.double().cpu().requires_grad_(), gamma.double().cpu().requires_grad_(), beta.double().cpu().requires_grad_()
    y = F.layer_norm(xa, (C_,), wa, ba, 1e-5)
    y.backward(gy.double().cpu())
    tol = 2e-5 * max(1.0, (N / 4096) ** 0.5)
    for got, ref, name in zip(outs[0], (y, xa.grad, wa.grad, ba.grad), ("y", "dx", "dgamma", "dbeta")):
        assert rel_err(got.numpy(), ref.detach().numpy()) < tol, (case, name)
    with pytest.raises(ValueError):
        train.layer_norm(x[:, :14].contiguous(), gamma[:14], beta[:14])      # C = 14: not a multiple of 4, not taken
    ln = nn.LayerNorm(C_).to(DEV)
    train.accelerate(ln)
    assert getattr(ln.forward, "__func__", None) is train._hip_layer_norm_forward


@pytest.mark.parametrize("case", sorted(GN_CASES))
@pytest.mark.parametrize("modulated", [False, True])
def test_group_norm_silu_fused_matches_a_float64_reference(case, modulated):
    """nd_groupnorm_silu_train_forward / _backward == silu(group_norm(x) * (scale + 1) + shift) and its autograd in float64:
    y, dx, dgamma, dbeta and d(scale | shift); bitwise repeatable."""
    B, C_, G, H, W = GN_CASES[case]
    x = (U(case + ".x", (B, C_, H, W), -1.5, 1.5) + 3.0 * U(case + ".m", (B, C_, 1, 1))).to(DEV)
    gamma, beta = U(case + ".g", (C_,), 0.5, 1.5).to(DEV), U(case + ".b", (C_,)).to(DEV)
    ss = U(case + ".ss", (B, 2 * C_, 1, 1), -0.5, 0.5).to(DEV) if modulated else None
    gy = U(case + ".gy", (B, C_, H, W)).to(DEV)
    outs = []
    for _ in range(2):
        xa, wa, ba = x.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
        sa = ss.clone().requires_grad_() if modulated else None
        y = train.group_norm_silu(xa, G, wa, ba, sa, 1e-5)
        y.backward(gy)
        outs.append([t.detach().float().cpu().contiguous() for t in (y, xa.grad, wa.grad, ba.grad) + ((sa.grad,) if modulated else ())])
    assert all(torch.equal(p, q) for p, q in zip(*outs))
    xa, wa, ba = x.double().cpu().requires_grad_(), gamma.double().cpu().requires_grad_(), beta.double().cpu().requires_grad_()
    n = F.group_norm(xa, G, wa, ba, 1e-5)
    sa = None
    if modulated:
        sa = ss.double().cpu().requires_grad_()
        scale, shift = sa.chunk(2, dim=1)
        n = n * (scale + 1) + shift
    y = F.silu(n)
    y.backward(gy.double().cpu())
    refs = (y, xa.grad, wa.grad, ba.grad) + ((sa.grad,) if modulated else ())
    for got, ref, name in zip(outs[0], refs, ("y", "dx", "dgamma", "dbeta", "dscale_shift")):
        assert got.shape == ref.shape, name
        assert rel_err(got.numpy(), ref.detach().numpy()) < 3e-5, (case, name)


class Block(nn.Module):
    """The reference's Block by shape (Diffusion_arch.py:128-144): what accelerate() recognises structurally."""

    def __init__(self, cin, cout, groups=8):
        super().__init__()
        self.proj, self.norm, self.act = nn.Conv2d(cin, cout, 3, padding=1), nn.GroupNorm(groups, cout), nn.SiLU()

    def forward(self, x, scale_shift=None):
        x = self.norm(self.proj(x))
        if scale_shift is not None:
            scale, shift = scale_shift
            x = x * (scale + 1) + shift
        return self.act(x)


def test_accelerate_fuses_the_tail_of_block_shaped_modules():
    torch.manual_seed(3)
    ref = Block(32, 64).to(DEV)
    hip = copy.deepcopy(ref)
    train.accelerate(hip)
    assert getattr(hip.forward, "__func__", None) is train._hip_block_forward
    x = U("bk.x", (2, 32, 24, 40)).to(DEV)
    e = U("bk.e", (2, 128, 1, 1), -0.5, 0.5).to(DEV)
    maps = U("bk.m", (2, 128, 24, 40), -0.5, 0.5).to(DEV)
    for ss_src in (None, e, maps):                         # plain, the time embedding's per-sample modulation (fused), per-pixel maps (unfused path)
        res = []
        for net in (ref, hip):
            net.zero_grad(set_to_none=True)
            xa = x.clone().requires_grad_()
            sa = None if ss_src is None else ss_src.clone().requires_grad_()
            y = net(xa, None if sa is None else sa.chunk(2, dim=1))
            y.square().mean().backward()
            res.append([y.detach().cpu(), xa.grad.cpu()] + [p.grad.detach().cpu() for p in net.parameters()] + ([sa.grad.cpu()] if sa is not None else []))
        for got, want in zip(res[1], res[0]):
            assert rel_err(got.numpy(), want.numpy()) < 2e-4


@pytest.mark.parametrize("weight_decay", [0.0, 0.01])
def test_adam_one_launch_step_matches_torch_optim_adam(weight_decay):
    """train.Adam (nd_adam_step_f32: all parameters of a group in one launch) against torch.optim.Adam over several steps on parameters of awkward
    sizes (one element, not a multiple of 4, several chunks, an unaligned view-backed tensor), with a parameter that has no gradient, a state dict
    moved from one class to the other in the middle, and bitwise repeatability."""
    shapes = [(1,), (7,), (3, 5, 3, 3), (64, 64, 3, 3), (20000,), (130, 257)]

    def make():
        torch.manual_seed(3)
        ps = [nn.Parameter(torch.randn(*s, device=DEV)) for s in shapes]
        big = torch.randn(4099, device=DEV)
        ps.append(nn.Parameter(big[1:4098].clone()))                  # (a fresh allocation: aligned; the vec4 path also needs whole chunks)
        ps.append(nn.Parameter(torch.randn(5, device=DEV)))           # never receives a gradient
        return ps

    def grads(ps, step):
        g = torch.Generator(device="cpu").manual_seed(100 + step)
        for p in ps[:-1]:
            p.grad = (torch.randn(p.shape, generator=g) * (0.5 + step)).to(DEV)

    runs = []
    for cls in (torch.optim.Adam, train.Adam, train.Adam):
        ps = make()
        opt = cls(ps, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=weight_decay)
        for step in range(4):
            grads(ps, step)
            opt.step()
            if step == 1:                                             # the state dict of one class loads into the other
                other = (train.Adam if cls is torch.optim.Adam else torch.optim.Adam)(ps, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=weight_decay)
                other.load_state_dict(opt.state_dict())
                opt = other
        torch.cuda.synchronize()
        runs.append(([p.detach().cpu() for p in ps], opt.state_dict()))
    ref, got, again = runs
    for a, b, c in zip(ref[0], got[0], again[0]):
        assert torch.equal(b, c)                                      # no reductions: the same bits
        assert rel_err(b.numpy(), a.numpy()) < 2e-6
    assert torch.equal(got[0][-1], make()[-1].detach().cpu())         # no gradient: untouched
    for k in ref[1]["state"]:
        for name in ("exp_avg", "exp_avg_sq"):
            assert rel_err(got[1]["state"][k][name].cpu().numpy(), ref[1]["state"][k][name].cpu().numpy()) < 2e-6
        assert float(got[1]["state"][k]["step"]) == float(ref[1]["state"][k]["step"]) == 4.0
    with pytest.raises(NotImplementedError):
        train.Adam(make(), amsgrad=True)
    with pytest.raises(L.HipError):
        L.call("nd_adam_step_f32", None, 1, None, 1, 0.9, 0.99, 1e-8, 0.0, None)


def test_training_with_the_one_launch_adam_follows_torch_adam_step_by_step():
    """The packed-operand caches (3x3 weights on F(4x4), Linear / 1x1 weights, views of parameters) are keyed on version counters: an optimizer that
    writes through raw pointers must bump them, or every forward after the first step runs on the initial weights.  Five steps of a block-shaped net
    (3x3 convs wide enough for the F(4x4) kernels, a 1x1 conv, GroupNorm) with train.Adam against the same net under torch.optim.Adam."""
    torch.manual_seed(0)
    net = nn.Sequential(_Block(32, 64), _Block(64, 64), nn.Conv2d(64, 4, 1)).to(DEV)
    train.accelerate(net)
    x = U("adamnet.x", (2, 32, 32, 64)).to(DEV)
    target = U("adamnet.t", (2, 4, 32, 64)).to(DEV)
    curves = []
    for cls in (torch.optim.Adam, train.Adam):
        m = copy.deepcopy(net)
        opt = cls(m.parameters(), lr=2e-3)
        v0 = [p._version for p in m.parameters()]
        losses = []
        for _ in range(5):
            opt.zero_grad(set_to_none=True)
            loss = F.mse_loss(m(x), target)
            loss.backward()
            opt.step()
            losses.append(float(loss))
        assert all(p._version > v for p, v in zip(m.parameters(), v0))
        curves.append(losses)
    assert curves[0][-1] < 0.9 * curves[0][0]                                    # it does train
    assert curves[1] == pytest.approx(curves[0], rel=2e-4)


@pytest.mark.parametrize("form", [0, 2, 3])
@pytest.mark.parametrize("shape", [(2, 16, 32, 32, 64), (2, 16, 24, 32, 64)])
def test_wgrad_reads_channel_slices_of_wider_tensors(shape, form, wgrad_form):
    """x and dy as channel slices of wider NHWC tensors (pixel strides ldx > cin, ldy > cout: the halves of a concat, a gradient that is a view), both
    forms of the kernel (W = 32: Winograd domain, W = 24: nine taps), against the gradient computed from contiguous copies -- the same bits."""
    lib = L.load()
    wgrad_form(form)                                                 # the product's choice, four waves, eight waves (cout = 64)
    B, H, W, cin, cout = shape
    xw = U("wgs.x", (B, H, W, cin + 32)).to(DEV)
    gw = U("wgs.g", (B, H, W, cout + 16)).to(DEV)
    xs, gs = xw[..., 32:], gw[..., 8:8 + cout]                       # 16-byte aligned starts
    ws = torch.empty(int(lib.nd_conv3x3_wgrad_workspace_floats(B, H, W, cin, cout)), device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    res = []
    for x_, ldx, g_, ldy in ((xs, cin + 32, gs, cout + 16), (xs.contiguous(), cin, gs.contiguous(), cout)):
        dw = torch.full((cout, cin, 3, 3), float("nan"), device=DEV)
        db = torch.full((cout,), float("nan"), device=DEV)
        L.call("nd_conv3x3_wgrad_nhwc_f32", x_.data_ptr(), ldx, g_.data_ptr(), ldy, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), B, H, W, cin, cout, st)
        torch.cuda.synchronize()
        res.append((dw.cpu(), db.cpu()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    ref = torch.nn.grad.conv2d_weight(xs.permute(0, 3, 1, 2).double().cpu(), (cout, cin, 3, 3), gs.permute(0, 3, 1, 2).double().cpu(), padding=1)
    assert rel_err(res[0][0].numpy(), ref.numpy()) < 2e-5


@pytest.mark.parametrize("shape", [(2, 32, 64, 64, 64, 128), (4, 32, 32, 512, 256, 512), (1, 48, 96, 32, 64, 48), (1, 32, 32, 48, 16, 64), (4, 128, 128, 128, 64, 128)])
def test_convolutions_over_two_sources_equal_the_concatenation(shape):
    """train.conv3x3_cat / conv1x1_cat (the up path's cat((x, skip)) read as two kernel sources, never materialised): outputs, the GroupNorm statistics
    epilogue and every gradient equal those of the same operators on torch.cat -- bit for bit, since the kernels walk the same channel chunks in the same
    order (the deep case runs the split-K form); the 3x3 weight gradient reads its cin blocks from either source (nd_conv3x3_wgrad_cat_nhwc_f32: the same bits
    too) where both are whole 32-channel blocks."""
    B, H, W, c0, c1, cout = shape
    x0, x1 = U("cat.x0", (B, c0, H, W), -1.5, 1.5).to(DEV), U("cat.x1", (B, c1, H, W), -1.5, 1.5).to(DEV)
    w3 = (U("cat.w3", (cout, c0 + c1, 3, 3)) / (9 * (c0 + c1)) ** 0.5).to(DEV)
    w1 = (U("cat.w1", (cout, c0 + c1, 1, 1)) / (c0 + c1) ** 0.5).to(DEV)
    b = U("cat.b", (cout,)).to(DEV)
    gy = U("cat.gy", (B, cout, H, W)).to(DEV)
    assert train.cat_sources_ok(x0, x1)
    for kind, w in (("3x3", w3), ("1x1", w1)):
        outs = []
        for two in (False, True):
            a, c, wa, ba = x0.clone().requires_grad_(), x1.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
            if kind == "3x3":
                y, (st, sc) = train.conv3x3_cat(a, c, wa, ba, with_stats=True) if two else train.conv3x3_with_stats(torch.cat((a, c), 1), wa, ba)
            else:
                y, st = (train.conv1x1_cat(a, c, wa, ba) if two else train.conv1x1(torch.cat((a, c), 1), wa, ba)), None
            y.backward(gy)
            outs.append([t.detach().cpu() for t in (y, a.grad, c.grad, wa.grad, ba.grad)] + ([st.cpu()] if st is not None else []))
        names = ("y", "dx0", "dx1", "dw", "db", "stats")
        for got, ref, name in zip(outs[1], outs[0], names):
            if name in ("dw", "db") and (kind == "1x1" or c0 % 32 or c1 % 32):    # weight gradient per source: another split, another summation order
                assert rel_err(got.numpy(), ref.numpy()) < 2e-5, (kind, name)
            else:
                assert torch.equal(got, ref), (kind, name)
    assert not train.cat_sources_ok(x0[:, :8], x1)                                # half a 16-channel chunk: the caller concatenates


@pytest.mark.parametrize("shape", [(2, 32, 64, 64, 64, 64), (4, 32, 32, 512, 256, 512), (1, 48, 96, 32, 64, 48)])
@pytest.mark.parametrize("joined", [True, False])
def test_shortcut_of_a_concatenation_takes_the_block_convolutions_gradient_in_its_epilogue(shape, joined):
    """train.conv1x1_shortcut_cat: (x0, x1, res_conv(cat(x0, x1))) with the inputs handed on to the block's first convolution -- its data gradient (the two channel
    slices of ONE tensor out of conv3x3_cat: ``joined``) is the residual of the shortcut's own data-gradient GEMM instead of two autograd additions; any other
    consumer of the hand-over (``not joined``: the gradients arrive as separate tensors) takes the additions.  Every gradient against the unfused operators."""
    B, H, W, c0, c1, cout = shape
    x0, x1 = U("sc.x0", (B, c0, H, W), -1.5, 1.5).to(DEV), U("sc.x1", (B, c1, H, W), -1.5, 1.5).to(DEV)
    x0, x1 = x0.contiguous(memory_format=torch.channels_last), x1.contiguous(memory_format=torch.channels_last)
    w3 = (U("sc.w3", (cout, c0 + c1, 3, 3)) / (9 * (c0 + c1)) ** 0.5).to(DEV)
    w1 = (U("sc.w1", (cout, c0 + c1, 1, 1)) / (c0 + c1) ** 0.5).to(DEV)
    b = U("sc.b", (cout,)).to(DEV)
    gh, gr = U("sc.gh", (B, cout, H, W)).to(DEV), U("sc.gr", (B, cout, H, W)).to(DEV)
    outs = []
    for fused in (False, True):
        a, c, w3a, w1a, ba = (t.clone().requires_grad_() for t in (x0, x1, w3, w1, b))
        if fused:
            a2, c2, r = train.conv1x1_shortcut_cat(a, c, w1a, ba)
        else:
            a2, c2, r = a, c, train.conv1x1_cat(a, c, w1a, ba)
        h = train.conv3x3_cat(a2, c2, w3a, None) if joined else train.conv3x3(a2 * 1.5, w3a[:, :c0].contiguous(), None) + (c2 * c2).sum() * 1e-3
        torch.autograd.backward((h, r), (gh, gr))
        outs.append([t.detach().cpu() for t in (h, r, a.grad, c.grad, w3a.grad, w1a.grad, ba.grad)])
    for got, ref, name in zip(outs[1], outs[0], ("h", "r", "dx0", "dx1", "dw3", "dw1", "db")):
        if name in ("h", "r", "dw3", "dw1", "db"):
            assert torch.equal(got, ref), name                 # the same launches on the same operands
        else:
            assert rel_err(got.numpy(), ref.numpy()) < 2e-6, name          # (a + b in the epilogue against b + a by autograd: fp32 addition commutes, fma contraction may differ)


def test_a_training_step_with_the_capturable_adam_replays_as_one_graph():
    """train.Adam(capturable=True): step counters on the device, nothing computed on the host -- forward + backward + optimizer captured into one
    torch.cuda.CUDAGraph and replayed; the losses follow torch.optim.Adam run eagerly on a copy of the net (the packing caches' repack launches are part of
    the graph)."""
    torch.manual_seed(0)
    net = nn.Sequential(_Block(32, 64), _Block(64, 64), nn.Conv2d(64, 4, 1)).to(DEV)
    train.accelerate(net)
    x = U("capnet.x", (2, 32, 32, 64)).to(DEV)
    target = U("capnet.t", (2, 4, 32, 64)).to(DEV)
    ref, cap = copy.deepcopy(net), copy.deepcopy(net)
    opt_ref = torch.optim.Adam(ref.parameters(), lr=2e-3)
    want = []
    for _ in range(6):
        opt_ref.zero_grad(set_to_none=True)
        loss = F.mse_loss(ref(x), target)
        loss.backward()
        opt_ref.step()
        want.append(float(loss))
    opt = train.Adam(cap.parameters(), lr=2e-3, capturable=True)
    got = []
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):                                                   # two eager steps on a side stream (state, chunk tables), as torch's recipe asks
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            loss = F.mse_loss(cap(x), target)
            loss.backward()
            opt.step()
            got.append(float(loss))
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        gloss = F.mse_loss(cap(x), target)
        gloss.backward()
        opt.step()
    for _ in range(4):
        g.replay()
        torch.cuda.synchronize()
        got.append(float(gloss))
    assert float(opt.state[next(iter(cap.parameters()))]["step"]) == 6.0          # (the capture itself does not execute)
    assert got == pytest.approx(want, rel=3e-4)
    with pytest.raises(RuntimeError):
        bad = train.Adam(copy.deepcopy(net).parameters(), lr=1e-3)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            bad.step()


def test_packing_cache_sees_data_writes_after_invalidate_and_drops_dead_models():
    """the packing caches compare version counters, which ``p.data.copy_`` (ema_pytorch's shadow update, SID_arch's re-initialisation) does not
    bump.  Documented contract: train.invalidate_packs() after such writes.  Also: only nn.Parameters (and views of them) are cached, and a deleted
    model's entries are released with train.release_packs()."""
    torch.manual_seed(0)
    net = nn.Sequential(_Block(32, 64), nn.Conv2d(64, 64, 1)).to(DEV)
    train.accelerate(net)
    x = U("pc.x", (2, 32, 32, 32)).to(DEV)
    with torch.no_grad():
        y0 = net(x).clone()
        for p in net.parameters():
            p.data.mul_(0.5)                               # no version bump
        stale = net(x).clone()                             # documented hazard: still the old packings (where a packing is cached at all)
        train.invalidate_packs()
        y1 = net(x).clone()
    ref = copy.deepcopy(net).to(DEV)
    with torch.no_grad():
        want = ref(x)                                      # fresh module, fresh packings: the halved weights
    assert torch.allclose(y1, want, rtol=1e-5, atol=1e-6)
    assert not torch.allclose(y0, y1, rtol=1e-3, atol=1e-4)
    del stale
    cache = train._pack_cache(DEV)
    n_live = len(cache.entries)
    assert n_live > 0 and all(isinstance(e[cache._REF](), nn.Parameter) for e in cache.entries.values())
    # a non-parameter leaf is never cached
    w = torch.randn(64, 64, device=DEV)
    assert not train._PackCache.cacheable(w) and train._PackCache.cacheable(net[1].weight.flatten(1))
    # the cache holds its parameters until train.release_packs() (explicit, as documented): checked on a cache of its own, the suite's stays
    saved = dict(train._PACK_CACHES)
    try:
        train._PACK_CACHES.clear()
        small = nn.Conv2d(64, 64, 1).to(DEV)
        train.accelerate(small)
        with torch.no_grad():
            small(torch.randn(1, 64, 16, 16, device=DEV))
        assert len(train._pack_cache(DEV).entries) >= 1
        train.release_packs()
        assert not train._PACK_CACHES
    finally:
        train._PACK_CACHES.clear()
        train._PACK_CACHES.update(saved)


def test_adam_keeps_per_parameter_step_counts_and_updates_each_group_once():
    """(a) parameters stepped under different sets of gradients carry different step counts, and each must get its own bias correction;
    (b) a table rebuild (replaced optimizer state) must not re-run groups that were already updated in the same call; (c) the closure's loss is returned."""
    torch.manual_seed(0)
    pa, pb = [nn.Parameter(torch.randn(64, 32, device=DEV)) for _ in range(2)]
    qa, qb = [nn.Parameter(p.detach().clone()) for p in (pa, pb)]
    ours = train.Adam([{"params": [pa]}, {"params": [pb]}], lr=1e-2)
    theirs = torch.optim.Adam([{"params": [qa]}, {"params": [qb]}], lr=1e-2)
    g = [torch.randn(64, 32, device=DEV) for _ in range(6)]
    for i in range(6):
        for opt, (a, b) in ((ours, (pa, pb)), (theirs, (qa, qb))):
            a.grad = g[i].clone()
            b.grad = g[i].clone() * 0.5 if i % 2 == 0 else None          # the second group sits out every other step
            opt.step()
    assert torch.allclose(pa, qa, rtol=2e-5, atol=2e-6) and torch.allclose(pb, qb, rtol=2e-5, atol=2e-6)
    assert float(ours.state[pb]["step"]) == 3.0 and float(ours.state[pa]["step"]) == 6.0
    # same group, parameters with DIFFERENT counts: one group holding both, the second parameter joins late
    pc, pd = [nn.Parameter(torch.randn(32, 32, device=DEV)) for _ in range(2)]
    qc, qd = [nn.Parameter(p.detach().clone()) for p in (pc, pd)]
    ours2, theirs2 = train.Adam([pc, pd], lr=1e-2), torch.optim.Adam([qc, qd], lr=1e-2)
    for i in range(5):
        for opt, (c, d) in ((ours2, (pc, pd)), (theirs2, (qc, qd))):
            c.grad = g[i][:32].clone()
            d.grad = g[i][32:].clone() if i >= 2 else None
            opt.step()
    assert torch.allclose(pc, qc, rtol=2e-5, atol=2e-6) and torch.allclose(pd, qd, rtol=2e-5, atol=2e-6)
    # (b) + (c): replace the state between steps (load_state_dict), then one step with a closure
    sd = copy.deepcopy(ours.state_dict())
    ours.load_state_dict(sd)
    theirs.load_state_dict(copy.deepcopy(theirs.state_dict()))
    for opt, (a, b) in ((ours, (pa, pb)), (theirs, (qa, qb))):
        a.grad, b.grad = g[0].clone(), g[1].clone()
    before = float(ours.state[pa]["step"])
    out = ours.step(lambda: torch.tensor(3.5))
    theirs.step()
    assert float(out) == 3.5
    assert float(ours.state[pa]["step"]) == before + 1.0                 # once, not twice
    assert torch.allclose(pa, qa, rtol=2e-5, atol=2e-6) and torch.allclose(pb, qb, rtol=2e-5, atol=2e-6)
