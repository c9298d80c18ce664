"""GPU parity of the whole hot path: NoiseDiffNet.forward and GaussianDiffusion.sample on HIP
against the fixtures captured from the reference (tests/golden) and against the CPU oracle.

north-star tolerance: 1e-3 relative fp32.  Bounds used here are tighter where the measured
agreement allows; every bound is max|a-b| / max(1, max|ref|).
"""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from noisediff_amd import GaussianDiffusion, NoiseDiffNet, synth
from oracle import noisediff_oracle as O
from util import close, noise_fn, rel_err, state_dict, sub

DEV = torch.device("cuda", 0)
NET_TOL = 2e-4
SAMPLE_TOL = 1e-3


def make_net(dim, mid_attn=False):
    args = SimpleNamespace(dim=dim, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False, mid_attn=mid_attn)
    net = NoiseDiffNet(args)
    net.load_state_dict(state_dict(dim, mid_attn=mid_attn), strict=True)
    return net.to(DEV).eval()


def to_dev(cond):
    # the reference leaves iso_ratio_idx on the CPU (trainer_diffusion.py:135)
    return {k: (v if k == "iso_ratio_idx" else v.to(DEV)) for k, v in cond.items()}


@pytest.mark.parametrize("dim,H", [(16, 32), (32, 64)])
def test_net_forward_matches_reference_golden(golden, dim, H):
    B = 2
    net = make_net(dim)
    cond = synth.make_condition(B, H, seed=1)
    x = synth.make_noise(2, "net.x", B, 4, H)
    with torch.inference_mode():
        for t in (0, 500, 999):
            y = net(x.to(DEV), torch.full((B,), t, dtype=torch.long, device=DEV), to_dev(cond))
            assert y.shape == (B, 4, H, H) and y.device.type == "cuda"
            assert close(y.cpu().numpy(), golden("net", f"net.d{dim}.h{H}.t{t}"), NET_TOL), t
        y = net(x.to(DEV), torch.tensor([3, 777], device=DEV), to_dev(cond))
        assert close(y.cpu().numpy(), golden("net", f"net.d{dim}.h{H}.tmixed"), NET_TOL)


@pytest.mark.parametrize("dim,H", [(48, 64), (64, 32)])
def test_net_forward_other_widths_match_oracle(dim, H):
    """Widths the fixtures do not cover: 48 (the reference's default --dim: channel counts that are not multiples of 32,
    K chunks straddling concat sources -> the fallback kernels) and 64 (the benchmark width), against the pinned oracle."""
    B = 2
    net = make_net(dim)
    sd = state_dict(dim)
    cond = synth.make_condition(B, H, seed=3)
    x = synth.make_noise(4, "net.x", B, 4, H)
    t = torch.tensor([17, 640])
    with torch.inference_mode():
        y = net(x.to(DEV), t.to(DEV), to_dev(cond)).cpu()
        ref = O.noisediff_forward(sd, x, t, cond)
    assert close(y.numpy(), ref.numpy(), NET_TOL)


def test_net_intermediates_match_reference_taps(golden):
    """Layer-by-layer agreement (debug plan keeps every named intermediate)."""
    dim, H, B = 32, 64, 2
    net = make_net(dim)
    plan = net.hip_engine(DEV).plan(B, H, H, debug=True)
    plan.set_condition(to_dev(synth.make_condition(B, H, seed=1)))
    plan.forward(synth.make_noise(2, "net.x", B, 4, H).to(DEV), torch.full((B,), 500, dtype=torch.long))
    for name in ("pos_block1", "down0", "down1", "down2", "down3", "mid", "up0", "up1", "up2", "up3", "shot_noise"):
        t = plan.taps[name]
        got = t.view(B, -1, t.shape[-1]).permute(0, 2, 1).contiguous().cpu()     # NHWC -> NCHW flattening
        assert close(sub(got), golden("net", f"net.d{dim}.h{H}.t500.tap.{name}"), NET_TOL), name
    pe = plan.pos_emb.view(B, -1, 8).permute(0, 2, 1).contiguous().cpu()
    assert rel_err(sub(pe), golden("net", f"net.d{dim}.h{H}.t500.tap.pos_emb")) < 1e-5


def test_net_forward_non_square_and_wrapped(golden):
    """H != W, ragged tiles (40x24) and the DataParallel-wrapped form, against the oracle."""
    dim, B, H, W = 16, 3, 40, 24
    sd = state_dict(dim)
    net = make_net(dim)
    cond = {"clean_img": synth.uniform(5, "ns.clean", (B, 4, H, W), 0, 1),
            "position": synth.uniform(5, "ns.pos", (B, 2, H, W), 0, 1),
            "iso_ratio_idx": synth.randint(5, "ns.iso", (B,), 0, 75)}
    x = synth.uniform(5, "ns.x", (B, 4, H, W), -2, 2)
    t = torch.tensor([0, 123, 999])
    with torch.no_grad():
        ref = O.noisediff_forward(sd, x, t, cond)
        got = torch.nn.DataParallel(net, device_ids=[0])(x.to(DEV), t.to(DEV), to_dev(cond))
    assert close(got.cpu().numpy(), ref.numpy(), NET_TOL)


def test_net_refuses_cpu_and_every_architecture_runs_under_autograd():
    from noisediff_amd._lib import HipError
    from noisediff_amd import UNet_PosEmbV2_NoPosition
    net = make_net(16)
    cond = synth.make_condition(1, 16, seed=1)
    x = torch.zeros(1, 4, 16, 16)
    with torch.no_grad(), pytest.raises(HipError):
        net(x, torch.zeros(1, dtype=torch.long), cond)                   # CPU tensor: no fallback
    with pytest.raises(HipError):                                        # ... under autograd neither
        net.cpu()(x, torch.zeros(1, dtype=torch.long), cond)
    # autograd requested: since r5 every architecture has the differentiable path (tests/test_trainable.py pins loss and gradients to the reference's)
    y = make_net(16, mid_attn=True)(x.to(DEV), torch.zeros(1, dtype=torch.long, device=DEV), to_dev(cond))
    assert y.requires_grad and y.shape == (1, 4, 16, 16)
    v = UNet_PosEmbV2_NoPosition(SimpleNamespace(dim=16, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False)).to(DEV)
    y = v(x.to(DEV), torch.zeros(1, dtype=torch.long, device=DEV), x.to(DEV))
    assert y.requires_grad and y.shape == (1, 4, 16, 16)


def _sample(dim, B, H, T, S, eta=0.0, return_all=False, sched="sigmoid2", objective="pred_v", mid=False, preset=False):
    net = make_net(dim, mid_attn=mid)
    gd = GaussianDiffusion(torch.nn.DataParallel(net, device_ids=[0]), image_size=H, timesteps=T, sampling_timesteps=S,
                           beta_schedule=sched, objective=objective, ddim_sampling_eta=eta).to(DEV)
    n_draws = (S if S is not None else T) - 1
    noise = {"steps": torch.stack([synth.make_noise(2, f"noise.{i}", B, 4, H) for i in range(n_draws)])}
    x_T = synth.make_noise(2, "x_T", B, 4, H)
    kw = {"preset_mean": x_T.to(DEV)} if preset else {}
    if not preset:
        noise["x_T"] = x_T
    out = gd.sample(batch_size=B, condition=to_dev(synth.make_condition(B, H, seed=1)), return_all_timesteps=return_all,
                    noise=noise, **kw)
    assert out.device.type == "cuda" and out.dtype == torch.float32
    return out.cpu().numpy()


def test_sampler_config1_ddim50(golden):
    """BASELINE config 1 end to end: d=32, 64x64x4, 50-step DDIM, batch 4."""
    assert close(_sample(32, 4, 64, 1000, 50), golden("sampler", "samp.cfg1.out"), SAMPLE_TOL)


def test_sampler_ddpm20_and_preset_mean(golden):
    assert close(_sample(16, 2, 32, 20, None), golden("sampler", "samp.ddpm20.out"), SAMPLE_TOL)
    assert close(_sample(16, 2, 32, 20, None, preset=True), golden("sampler", "samp.ddpm20_preset.out"), SAMPLE_TOL)


def test_sampler_return_all_and_eta(golden):
    res = _sample(16, 2, 32, 4, None, return_all=True)
    assert res.shape == (2, 5, 4, 32, 32)
    assert close(res, golden("sampler", "samp.ddpm4_all.out"), SAMPLE_TOL)
    res = _sample(16, 2, 32, 20, 5, eta=0.5, return_all=True)
    assert res.shape == (2, 6, 4, 32, 32)
    assert close(res, golden("sampler", "samp.ddim5_eta.out"), SAMPLE_TOL)


def test_sampler_other_objectives(golden):
    assert close(_sample(16, 2, 32, 50, None, sched="linear", objective="pred_noise"), golden("sampler", "samp.ddpm50_eps_linear.out"), SAMPLE_TOL)
    assert close(_sample(16, 2, 32, 20, 5, sched="cosine", objective="pred_x0"), golden("sampler", "samp.ddim5_x0_cosine.out"), SAMPLE_TOL)


def test_sampler_config4_toy_mid_attention(golden):
    assert close(_sample(16, 2, 64, 1000, 10, mid=True), golden("sampler", "samp.cfg4toy.out"), SAMPLE_TOL)


def test_pre_clamp_model_output_matches(golden):
    """With random weights many final pixels saturate at the +-1 clamp; check the un-clamped v too."""
    dim, B, H = 32, 4, 64
    net = make_net(dim)
    cond = to_dev(synth.make_condition(B, H, seed=1))
    x = synth.make_noise(2, "x_T", B, 4, H)
    with torch.inference_mode():
        v0 = net(x.to(DEV), torch.full((B,), 999, dtype=torch.long), cond)
    assert close(v0.cpu().numpy(), golden("sampler", "samp.cfg1.v0"), NET_TOL)


def test_device_noise_mode_properties():
    """Throughput mode (Philox): repeatable for a seed, different across seeds, shard-invariant."""
    dim, B, H = 16, 4, 32
    net = make_net(dim)
    gd = GaussianDiffusion(net, image_size=H, timesteps=50, sampling_timesteps=None, beta_schedule="sigmoid2").to(DEV)
    cond = synth.make_condition(B, H, seed=1)
    a = gd.sample(batch_size=B, condition=to_dev(cond), seed=7).cpu()
    b = gd.sample(batch_size=B, condition=to_dev(cond), seed=7).cpu()
    c = gd.sample(batch_size=B, condition=to_dev(cond), seed=8).cpu()
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert torch.isfinite(a).all() and float(a.abs().max()) <= 1.0 + 1e-6      # last step returns clamped mean
    # rank shard [2, 4) of the same global batch reproduces rows 2..3 (no collective needed)
    gd.sample_offset = 2
    half = {k: v[2:] for k, v in cond.items()}
    s = gd.sample(batch_size=2, condition=to_dev(half), seed=7).cpu()
    gd.sample_offset = 0
    assert rel_err(s.numpy(), a[2:].numpy()) < 1e-5
    torch.manual_seed(3)
    d1 = gd.sample(batch_size=B, condition=to_dev(cond)).cpu()
    torch.manual_seed(3)
    d2 = gd.sample(batch_size=B, condition=to_dev(cond)).cpu()
    assert torch.equal(d1, d2)


def test_graph_replay_equals_eager_launches():
    dim, B, H = 16, 2, 32
    net = make_net(dim)
    gd = GaussianDiffusion(net, image_size=H, timesteps=30, beta_schedule="sigmoid2").to(DEV)
    cond = to_dev(synth.make_condition(B, H, seed=1))
    ref = gd.sample(batch_size=B, condition=cond, seed=11).cpu()
    loop = next(iter(gd._loop_cache.values()))
    eager = loop.run(x_T=None, step_noise=None, seed=11, first_sample=0, use_graph=False).cpu()
    assert torch.equal(ref, eager)


def test_per_step_public_methods_agree_with_the_fused_loop():
    """p_sample (:366-373) driven step by step with the reference's draws == the fused device loop's trajectory (one step kernel + graph per step);
    p_sample_loop / ddim_sample == sample() of a wrapper configured that way, whatever THIS wrapper was configured with; model_predictions on the
    HIP network == the oracle's x_0 / eps at one timestep."""
    dim, B, H, T = 16, 2, 32, 6
    net = make_net(dim)
    sd = state_dict(dim)
    cond_cpu = synth.make_condition(B, H, seed=1)
    cond = to_dev(cond_cpu)
    gd = GaussianDiffusion(net, image_size=H, timesteps=T, beta_schedule="sigmoid2").to(DEV)
    x_T = synth.make_noise(2, "x_T", B, 4, H)
    steps = torch.stack([synth.make_noise(2, f"noise.{i}", B, 4, H) for i in range(T - 1)])
    traj = gd.sample(batch_size=B, condition=cond, return_all_timesteps=True, noise={"x_T": x_T, "steps": steps})
    img = x_T.to(DEV)
    for k, t in enumerate(reversed(range(T))):
        img, x0 = gd.p_sample(img, t, cond, noise=steps[k] if t > 0 else None)
        assert rel_err(img.cpu().numpy(), traj[:, k + 1].cpu().numpy()) < 1e-5, t
    tb = torch.full((B,), 3, dtype=torch.long, device=DEV)
    with torch.no_grad():
        mp = gd.model_predictions(x_T.to(DEV), tb, cond, clip_x_start=True, rederive_pred_noise=True)
    out = O.noisediff_forward(sd, x_T, tb.cpu(), cond_cpu)
    eps_ref, x0_ref = O.predict_x0_eps(O.schedule_buffers("sigmoid2", T), "pred_v", x_T, 3, out, clip=True)
    assert rel_err(mp.pred_x_start.cpu().numpy(), x0_ref.numpy()) < NET_TOL and rel_err(mp.pred_noise.cpu().numpy(), eps_ref.numpy()) < 10 * NET_TOL
    a = gd.p_sample_loop((B, 4, H, H), cond, seed=5)
    assert torch.equal(a, gd.sample(batch_size=B, condition=cond, seed=5))
    dd = GaussianDiffusion(net, image_size=H, timesteps=T, sampling_timesteps=3, beta_schedule="sigmoid2").to(DEV)
    assert torch.equal(dd.p_sample_loop((B, 4, H, H), cond, seed=5), a)                 # a DDIM wrapper asked for the DDPM chain
    assert torch.equal(dd.ddim_sample((B, 4, H, H), cond, seed=5), dd.sample(batch_size=B, condition=cond, seed=5))
    assert dd.is_ddim_sampling and not gd.is_ddim_sampling


def test_full_size_properties_config2():
    """BASELINE config 2 shape (d=64, 128x128x4, batch 16), a few DDPM steps: size-independent properties --
    per-sample independence (batch of 16 == the same samples run as batch of 4) and finiteness."""
    dim, B, H = 64, 16, 128
    net = make_net(dim)
    gd = GaussianDiffusion(net, image_size=H, timesteps=1000, sampling_timesteps=4, beta_schedule="sigmoid2").to(DEV)
    cond = synth.make_condition(B, H, seed=1)
    full = gd.sample(batch_size=B, condition=to_dev(cond), seed=5).cpu()
    assert torch.isfinite(full).all()
    gd.sample_offset = 8
    part = gd.sample(batch_size=4, condition=to_dev({k: v[8:12] for k, v in cond.items()}), seed=5).cpu()
    assert rel_err(part.numpy(), full[8:12].numpy()) < 1e-4


# --------------------------------------------------------------------------- next row 8f-1: LSID + config-5 composition

def make_lsid():
    from noisediff_amd import LSID
    from noisediff_amd.spec import lsid_param_spec
    net = LSID(SimpleNamespace())
    net.load_state_dict(synth.make_state_dict(lsid_param_spec(), 0), strict=True)
    return net.to(DEV).eval()


def test_lsid_forward_matches_reference_golden(golden):
    net = make_lsid()
    with torch.inference_mode():
        for (B, H, W) in ((2, 64, 64), (1, 36, 44)):          # second size: ceil-mode pooling + crop after ConvTranspose
            x = synth.uniform(9, f"lsid.x.{H}x{W}", (B, 4, H, W), 0.0, 1.0)
            y = net(x.to(DEV))
            assert y.shape == (B, 4, H, W)
            assert close(y.cpu().numpy(), golden("lsid", f"lsid.{H}x{W}"), NET_TOL)


def test_maxpool_and_conv_transpose_kernels():
    """nn.MaxPool2d(2, 2, ceil_mode=True) and ConvTranspose2d(2, stride=2) + crop through the C ABI."""
    import ctypes as C
    import torch.nn.functional as F
    import hiputil as hu
    from noisediff_amd import _lib as L
    ctx = hu.Ctx()
    x = synth.uniform(4, "mp.x", (2, 8, 9, 7), -1, 1)
    xd, out = hu.nhwc(x), hu.full((2, 5, 4, 8))
    L.call("nd_maxpool2x2_nhwc_f32", xd.data_ptr(), out.data_ptr(), 2, 9, 7, 8, ctx.stream)
    ctx.sync()
    assert torch.equal(hu.nchw(out), F.max_pool2d(x, 2, 2, 0, ceil_mode=True))
    xi = synth.uniform(4, "ct.x", (2, 16, 5, 6), -1, 1)
    w = synth.uniform(4, "ct.w", (16, 8, 2, 2), -0.3, 0.3)
    ref = F.conv_transpose2d(F.leaky_relu(xi, 0.2), w, stride=2)[:, :, :9, :11]          # cropped like SID_arch.py:135
    wp = hu.pack_pw(ctx, w.permute(2, 3, 1, 0).reshape(32, 16).contiguous())
    outp = hu.full((2, 9, 11, 8))
    d = L.Pointwise()
    xid = hu.nhwc(xi)
    d.src, d.weight, d.out = hu.src(xid, None, L.PRO_LEAKY), wp.data_ptr(), outp.data_ptr()
    d.B, d.HW, d.W, d.cin, d.cout, d.ldo = 2, 30, 6, 16, 32, 8
    d.shuffle_c, d.shuffle_h, d.shuffle_w = 8, 9, 11
    L.call("nd_pointwise_gemm_nhwc_f32", C.byref(d), ctx.stream)
    ctx.sync()
    assert rel_err(hu.nchw(outp), ref) < 2e-5


def test_config5_noise_synthesis_feeds_denoiser_psnr(golden):
    """End to end (BASELINE config 5, scaled down): sampled noise -> clip/compose -> LSID -> PSNR, HIP vs oracle;
    plus the composition fixture captured from the reference's LSID."""
    from noisediff_amd import io
    from noisediff_amd.spec import lsid_param_spec
    lsid = make_lsid()
    sd_l = synth.make_state_dict(lsid_param_spec(), 0)
    clean = synth.uniform(9, "lsid.clean", (2, 4, 64, 64), 0.0, 1.0)
    noise = synth.make_noise(9, "lsid.noise", 2, 4, 64) * 0.1
    with torch.inference_mode():
        den = lsid(io.compose_noisy(noise, clean).to(DEV)).clamp(0, 1).cpu()
    assert close(den.numpy(), golden("lsid", "lsid.compose.out"), NET_TOL)
    assert abs(io.psnr(den, clean) - float(golden("lsid", "lsid.compose.psnr"))) < 1e-3
    # sampler -> denoiser: 8-step DDIM noise patches from the HIP sampler, then the same chain on both sides
    dim, B, H = 16, 2, 64
    net = make_net(dim)
    gd = GaussianDiffusion(net, image_size=H, timesteps=1000, sampling_timesteps=8, beta_schedule="sigmoid2").to(DEV)
    cond = synth.make_condition(B, H, seed=1)
    x_T = synth.make_noise(2, "x_T", B, 4, H)
    steps = torch.stack([synth.make_noise(2, f"noise.{i}", B, 4, H) for i in range(7)])
    with torch.inference_mode():
        gen = gd.sample(batch_size=B, condition=to_dev(cond), noise={"x_T": x_T, "steps": steps})
        den = lsid(io.compose_noisy(gen, cond["clean_img"].to(DEV))).cpu()
    ref_gen = O.sample(state_dict(dim), cond, image_size=H, batch_size=B, timesteps=1000, sampling_timesteps=8, x_T=x_T, noise=lambda i, s: steps[i])
    _, ref_den, ref_psnr = O.compose_and_denoise(sd_l, ref_gen, cond["clean_img"])
    assert close(gen.cpu().numpy(), ref_gen.numpy(), SAMPLE_TOL)
    assert abs(io.psnr(den, cond["clean_img"]) - ref_psnr) < 1e-2


def test_condition_batch_mismatch_and_bad_iso_are_rejected():
    """Trainer.test passes batch_size=args.batch_size even for a short last batch (trainer_diffusion.py:286): the
    reference fails deep inside torch.cat; here the mismatch is reported up front.  Out-of-range ISO indices raise
    like nn.Embedding does."""
    net = make_net(16)
    gd = GaussianDiffusion(net, image_size=32, timesteps=4, beta_schedule="sigmoid2").to(DEV)
    cond = synth.make_condition(3, 32, seed=1)
    with pytest.raises(ValueError, match="do not match batch"):
        gd.sample(batch_size=4, condition=to_dev(cond))
    bad = dict(cond, iso_ratio_idx=torch.tensor([0, 100, 5]))
    with pytest.raises(IndexError):
        gd.sample(batch_size=3, condition=to_dev(bad))
    with pytest.raises(AssertionError, match="divisible by 8"):
        net.hip_engine(DEV).plan(1, 20, 32)
    out = gd.sample(batch_size=3, condition=to_dev(cond), seed=1)
    assert out.shape == (3, 4, 32, 32)


# --------------------------------------------------------------------------- the reference's single-process multi-GPU entry (nn.DataParallel)

def test_data_parallel_device_ids_shard_the_batch_in_sample():
    """define_G wraps the net in nn.DataParallel(net, gpu_ids) (models/modules.py:73-83).  With several device_ids GaussianDiffusion.sample
    splits the batch rows over those devices -- one engine / plan / step graph per shard, launched round-robin -- instead of running
    silently on one.  Two logical shards on cuda:0 (ragged 3 + 2 and even 2 + 2): equal to the one-device run bit for bit (device Philox
    keyed by the global sample index; batch-invariant kernel selection) and, with injected noise, to the oracle along the trajectory."""
    dim, H, T = 16, 32, 6
    net = make_net(dim)
    one = GaussianDiffusion(torch.nn.DataParallel(net, device_ids=[0]), image_size=H, timesteps=T, beta_schedule="sigmoid2").to(DEV)
    two = GaussianDiffusion(torch.nn.DataParallel(net, device_ids=[0, 0]), image_size=H, timesteps=T, beta_schedule="sigmoid2").to(DEV)
    assert [d.index for d in two._sampling_devices()] == [0, 0] and len(one._sampling_devices()) == 1
    for B in (5, 4):
        cond = synth.make_condition(B, H, seed=1)
        a = one.sample(batch_size=B, condition=to_dev(cond), seed=11)
        b = two.sample(batch_size=B, condition=to_dev(cond), seed=11)
        assert b.shape == (B, 4, H, H) and b.device == a.device
        assert torch.equal(a, b)
    B = 5
    cond = synth.make_condition(B, H, seed=1)
    x_T = synth.make_noise(2, "x_T", B, 4, H)
    steps = torch.stack([synth.make_noise(2, f"noise.{i}", B, 4, H) for i in range(T - 1)])
    traj = two.sample(batch_size=B, condition=to_dev(cond), return_all_timesteps=True, noise={"x_T": x_T, "steps": steps}).cpu()
    ref = O.sample(state_dict(dim), cond, image_size=H, batch_size=B, timesteps=T, x_T=x_T, noise=lambda i, s: steps[i], return_all=True)
    assert traj.shape == ref.shape == (B, T + 1, 4, H, H)
    assert close(traj.numpy(), ref.numpy(), SAMPLE_TOL)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the pool's boxes have one; tools/first_node_check.sh runs this on a node)")
def test_data_parallel_over_two_physical_devices_equals_one_device():
    """The same comparison with device_ids=[0, 1]: the second shard's engine is a device-to-device copy of the first's arena, its step graph is
    captured and launched on cuda:1's stream from the one host thread (device guard in _Loop.advance / nd_graph_launch), and the gather copies
    its rows back to cuda:0.  Equal to the one-device run bit for bit."""
    dim, H, T, B = 16, 32, 6, 5
    net = make_net(dim)
    one = GaussianDiffusion(torch.nn.DataParallel(net, device_ids=[0]), image_size=H, timesteps=T, beta_schedule="sigmoid2").to(DEV)
    two = GaussianDiffusion(torch.nn.DataParallel(net, device_ids=[0, 1]), image_size=H, timesteps=T, beta_schedule="sigmoid2").to(DEV)
    assert [d.index for d in two._sampling_devices()] == [0, 1]
    cond = synth.make_condition(B, H, seed=1)
    a = one.sample(batch_size=B, condition=to_dev(cond), seed=11)
    b = two.sample(batch_size=B, condition=to_dev(cond), seed=11)
    assert b.device == a.device and torch.equal(a, b)
    assert torch.cuda.current_device() == DEV.index                          # the guards restore the caller's device


def test_graph_calls_make_the_streams_device_current():
    """nd_stream_device reports the device a library stream lives on; nd_graph_begin / _end / _launch run with that device current whatever
    the thread's current device is (one GPU here: the call sequence and the restored current device)."""
    import ctypes as C
    from noisediff_amd import _lib as L
    lib = L.load()
    st, g = C.c_void_p(), C.c_void_p()
    with torch.cuda.device(DEV):
        L.call("nd_stream_create", C.byref(st))
    assert lib.nd_stream_device(st) == DEV.index and lib.nd_stream_device(None) == -1
    x = torch.zeros(64, device=DEV)
    torch.cuda.synchronize()
    L.call("nd_graph_begin", st)
    L.call("nd_philox_normal_f32", x.data_ptr(), C.c_uint64(3), 0, -1, 1, 16, 4, st)
    L.call("nd_graph_end", st, C.byref(g))
    L.call("nd_graph_launch", g, st)
    L.call("nd_stream_sync", st)
    assert float(x.abs().sum()) > 0 and torch.cuda.current_device() == DEV.index
    L.call("nd_graph_destroy", g)
    L.call("nd_stream_destroy", st)


def test_data_parallel_forward_replicas_use_the_owners_engines():
    """nn.DataParallel.forward with several device_ids replicates the module (replicas have no parameters of their own) and calls the
    replicas from threads: they must find the owning module's packed weights.  Same result as the bare module."""
    dim, B, H = 16, 4, 32
    net = make_net(dim)
    dp = torch.nn.DataParallel(net, device_ids=[0, 0])
    cond = synth.make_condition(B, H, seed=3)
    x = synth.make_noise(4, "net.x", B, 4, H)
    t = torch.tensor([17, 640, 3, 999])
    with torch.inference_mode():
        ref = net(x.to(DEV), t.to(DEV), to_dev(cond))
        y = dp(x.to(DEV), t.to(DEV), {k: v.to(DEV) for k, v in cond.items()})
    assert y.shape == ref.shape and torch.equal(y, ref)


def test_low_latency_mode_split_k_for_small_batches_matches_oracle():
    """engine.SPLIT_K (ND_SPLIT_K=1): one patch at the headline size (d=64, 256x256) -- the 512-channel layers at 32x32 have 16 workgroup items --
    with the plain-source F(4x4) layers on the split-K form, statistics from its reduction kernel: the forward against the oracle and against the
    default path, and the recorded plan really splits."""
    from noisediff_amd import engine as E
    _oracle_threads_early()
    dim, H, B = 64, 256, 1
    sd = state_dict(dim)
    cond = synth.make_condition(B, H, seed=3)
    x = synth.make_noise(4, "net.x", B, 4, H)
    t = torch.tensor([640])
    base_net = make_net(dim)
    with torch.inference_mode():
        base = base_net(x.to(DEV), t.to(DEV), to_dev(cond)).cpu()
    old = E.SPLIT_K
    E.SPLIT_K = True
    try:
        net = make_net(dim)
        with torch.inference_mode():
            y = net(x.to(DEV), t.to(DEV), to_dev(cond)).cpu()
        plan = net.hip_engine(DEV).plan(B, H, H)
        split_ops = [m for _f, _a, name, m in plan.step_ops if name == "nd_conv3x3_wino4_splitk_nhwc_f32"]
    finally:
        E.SPLIT_K = old
    assert len(split_ops) >= 10 and max(m["splits"] for m in split_ops) == 8
    with torch.no_grad():
        ref = O.noisediff_forward(sd, x, t, cond)
    assert close(y.numpy(), ref.numpy(), NET_TOL)
    assert rel_err(y.numpy(), base.numpy()) < 1e-4 and not torch.equal(y, base)


def _oracle_threads_early():
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 8
    torch.set_num_threads(max(1, min(n, 32)))


# --------------------------------------------------------------------------- the bench workload's own sizes, against the oracle

def _oracle_threads():
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 8
    torch.set_num_threads(max(1, min(n, 32)))


@pytest.mark.parametrize("dim,H,B,mid", [(64, 256, 1, False), (64, 128, 2, False), (128, 64, 1, True), (128, 256, 1, True)])
def test_net_forward_headline_sizes_match_oracle(dim, H, B, mid):
    """One NoiseDiffNet.forward at the sizes the bench runs: d=64 at 256x256 (cfg3: every conv on the Winograd
    kernel, 512/768-channel layers on 32x32 images, the separate activation pass for cout >= 256), d=64 at 128x128 (cfg2), the
    cfg4 width d=128 with the mid Attention (channel counts 1024 / 1536 / 2048) at 64x64, and cfg4 AT ITS SIZE: d=128 + mid Attention at
    256x256 -- the F(4x4) kernel at 1024 / 1536 -> 1024 on 32x32 (64 / 96 K chunks), the large-tile 1x1 kernel at 1024 -> 2048 -> 1024
    and the attention kernel inside the net at N = 1024: the kernel selection of `bench.py --config cfg4`."""
    _oracle_threads()
    net = make_net(dim, mid_attn=mid)
    sd = state_dict(dim, mid_attn=mid)
    cond = synth.make_condition(B, H, seed=3)
    x = synth.make_noise(4, "net.x", B, 4, H)
    t = torch.tensor([640, 17][:B])
    with torch.inference_mode():
        y = net(x.to(DEV), t.to(DEV), to_dev(cond)).cpu()
        ref = O.noisediff_forward(sd, x, t, cond, mid_attention="mid_attn" if mid else None)
    assert float(ref.abs().max()) > 0.1
    assert close(y.numpy(), ref.numpy(), NET_TOL)


def test_config4_kernel_selection_at_its_size():
    """What `bench.py --config cfg4` launches (d=128 + mid Attention, 256x256): the 1024 / 1536 -> 1024 convs at 32x32 run on conv3x3_wino4,
    the wide token Linears are in the plan and the attention kernel sees N = 1024 -- so the parity cases above / below cover that selection."""
    net = make_net(128, mid_attn=True)
    plan = net.hip_engine(DEV).plan(1, 256, 256)
    by_layer = {m["layer"]: (name, m) for _, _, name, m in plan.step_ops if m}
    for layer, cin in (("mid_block1.block1.proj", 1024), ("ups.0.0.block1.proj", 1536), ("ups.0.1.block1.proj", 1536)):
        name, m = by_layer[layer]
        assert name == "nd_conv3x3_wino4_nhwc_f32" and (m["cin"], m["cout"], m["H"], m["W"]) == (cin, 1024, 32, 32), (layer, name, m)
    assert by_layer["downs.3.2.ff.net.0.0"][1]["cin"] == 512 and by_layer["downs.3.2.ff.net.0.0"][1]["cout"] == 1024
    assert by_layer["ups.0.2.ff.net.0.0"][1]["cin"] == 1024 and by_layer["ups.0.2.ff.net.0.0"][1]["cout"] == 2048
    assert by_layer["ups.0.2.ff.net.2"][1]["cin"] == 2048 and by_layer["ups.0.2.ff.net.2"][1]["cout"] == 1024
    assert any(name == "nd_attention_mfma_f32" and args[5] == 1024 for _, args, name, _ in plan.step_ops)


def test_sampler_config4_ddim8_at_256_matches_oracle():
    """BASELINE config 4 at its size -- d=128 with the mid-block Attention, 256x256x4, DDIM (8 of its 250 steps) -- against the oracle,
    with the un-clamped x_0 prediction path exercised by the trajectory (return_all_timesteps)."""
    _oracle_threads()
    dim, B, H, S = 128, 1, 256, 8
    net = make_net(dim, mid_attn=True)
    gd = GaussianDiffusion(net, image_size=H, timesteps=1000, sampling_timesteps=S, beta_schedule="sigmoid2").to(DEV)
    cond = synth.make_condition(B, H, seed=1)
    x_T = synth.make_noise(2, "x_T", B, 4, H)
    steps = torch.stack([synth.make_noise(2, f"noise.{i}", B, 4, H) for i in range(S - 1)])
    traj = gd.sample(batch_size=B, condition=to_dev(cond), return_all_timesteps=True, noise={"x_T": x_T, "steps": steps}).cpu()
    ref = O.sample(state_dict(dim, mid_attn=True), cond, image_size=H, batch_size=B, timesteps=1000, sampling_timesteps=S, x_T=x_T,
                   noise=lambda i, s: steps[i], return_all=True, mid_attention="mid_attn")
    assert traj.shape == ref.shape == (B, S + 1, 4, H, H)
    worst = max(rel_err(traj[:, k].numpy(), ref[:, k].numpy()) for k in range(S + 1))
    assert worst < SAMPLE_TOL, worst
    assert close(traj.numpy(), ref.numpy(), SAMPLE_TOL)                   # ... and every element of every x_t within 1e-5 + 1e-3 |ref|


def test_config4_batch8_at_256_matches_oracle_on_two_rows():
    """BASELINE config 4 per-GPU shard at its size (d=128 + mid Attention, 256x256x4, batch 8): a 2-step DDIM (eta 0.5) of the whole batch; rows 2 and 7 must
    equal the oracle run on those samples alone (parity at the batch the bench runs, per-sample independence)."""
    _oracle_threads()
    dim, B, H, S = 128, 8, 256, 2
    net = make_net(dim, mid_attn=True)
    gd = GaussianDiffusion(net, image_size=H, timesteps=1000, sampling_timesteps=S, ddim_sampling_eta=0.5, beta_schedule="sigmoid2").to(DEV)
    cond = synth.make_condition(B, H, seed=1)
    x_T = synth.make_noise(2, "x_T", B, 4, H)
    steps = torch.stack([synth.make_noise(2, f"noise.{i}", B, 4, H) for i in range(S - 1)])
    full = gd.sample(batch_size=B, condition=to_dev(cond), noise={"x_T": x_T, "steps": steps}).cpu()
    for row in (2, 7):
        one = {k: v[row:row + 1] for k, v in cond.items()}
        ref = O.sample(state_dict(dim, mid_attn=True), one, image_size=H, batch_size=1, timesteps=1000, sampling_timesteps=S, eta=0.5, x_T=x_T[row:row + 1],
                       noise=lambda i, s, r=row: steps[i][r:r + 1], mid_attention="mid_attn")
        assert close(full[row:row + 1].numpy(), ref.numpy(), SAMPLE_TOL / 10), row


def test_reference_shipped_workload_d48_at_512_forward_and_ddpm8_match_oracle():
    """The reference's own command line (script.sh:10: --dim 48 --crop_size 512 --batch_size 4, DDPM, sigmoid2, pred_v) at its size, one
    patch: a forward at t = 640 and a complete 8-step DDPM chain (every x_t) against the oracle.  cout = 48 / 96 pad to the
    F(4x4) kernel's 64-cout tile; d = 48 gives the stages 48 / 96 / 192 / 384 channels at 512 / 256 / 128 / 64 pixels."""
    _oracle_threads()
    dim, B, H = 48, 1, 512
    net = make_net(dim)
    sd = state_dict(dim)
    cond = synth.make_condition(B, H, seed=3)
    x = synth.make_noise(4, "net.x", B, 4, H)
    t = torch.tensor([640])
    with torch.inference_mode():
        y = net(x.to(DEV), t.to(DEV), to_dev(cond)).cpu()
        ref = O.noisediff_forward(sd, x, t, cond)
    assert float(ref.abs().max()) > 0.1
    assert close(y.numpy(), ref.numpy(), NET_TOL)
    T = 8                                                                # a complete 8-step DDPM chain (x_T -> x_0, the reference's draws injected), every x_t
    gd = GaussianDiffusion(net, image_size=H, timesteps=T, beta_schedule="sigmoid2").to(DEV)
    x_T = synth.make_noise(2, "x_T", B, 4, H)
    steps = torch.stack([synth.make_noise(2, f"noise.{i}", B, 4, H) for i in range(T - 1)])
    traj = gd.sample(batch_size=B, condition=to_dev(cond), return_all_timesteps=True, noise={"x_T": x_T, "steps": steps}).cpu()
    ref = O.sample(sd, cond, image_size=H, batch_size=B, timesteps=T, x_T=x_T, noise=lambda i, s_: steps[i], return_all=True)
    assert traj.shape == ref.shape == (B, T + 1, 4, H, H)
    worst = max(rel_err(traj[:, k].numpy(), ref[:, k].numpy()) for k in range(T + 1))
    assert worst < SAMPLE_TOL / 10, worst
    assert close(traj.numpy(), ref.numpy(), SAMPLE_TOL / 10)              # ... and every element of every x_t within 1e-5 + 1e-3 |ref|


def _conv_selection(plan):
    """(layer, entry point, tiling id, K ranges) of every 3x3 conv launch of a recorded plan, the pieces of one layer (a batch cut into several launches) collapsed."""
    sel = []
    for _, _, name, m in plan.step_ops:
        if m and "tiling" in m:
            item = (m["layer"], name, m["tiling"], m.get("splits", 1))
            if not sel or sel[-1] != item:
                sel.append(item)
    return sel


@pytest.mark.parametrize("dim,mid,batches", [(64, False, (1, 16, 64, 128)), (128, True, (1, 8, 32))])
def test_conv_kernel_selection_does_not_depend_on_the_batch(dim, mid, batches):
    """The reference takes any --batch_size (test_diffusion.py:23-78).  The F(4x4) kernels address a source through 32-bit offsets (below 1 GiB: d=64 at
    256 x 256 reaches that at 64 samples, d=128 at 32); the engine cuts a larger batch into equal pieces of whole samples for such a layer instead of falling
    to another kernel, so the kernel of every layer -- and with it a sample's bits -- is the same at every batch size.  Plans are recorded, not run."""
    net = make_net(dim, mid_attn=mid)
    eng = net.hip_engine(DEV)
    ref_sel = None
    for B in batches:
        plan = eng.plan(B, 256, 256)
        sel = _conv_selection(plan)
        assert len(sel) == 49 and all(name.startswith("nd_conv3x3_wino4") for _, name, _, _ in sel), (B, [s_ for s_ in sel if not s_[1].startswith("nd_conv3x3_wino4")])
        pieces = max(sum(1 for _, _, name, m in plan.step_ops if m and m.get("layer") == layer and "tiling" in m) for layer, *_ in sel)
        assert (pieces > 1) == (B * 256 * 256 * dim * 4 >= (1 << 30) - (1 << 16)), (B, pieces)      # the full-resolution layers are the first to be cut
        if ref_sel is None:
            ref_sel = sel
        assert sel == ref_sel, B
        del plan
        eng.plans.clear()                                                # (the workspace of 128 patches is ~50 GB)
        torch.cuda.empty_cache()


def test_rows_of_a_batch_of_64_equal_the_same_rows_in_a_batch_of_16_bit_for_bit():
    """d=64 at 256 x 256: at 64 samples the full-resolution convolutions run as two launches of 32 samples each; rows 16..31 of that batch and the same 16 samples
    as a batch of their own come out identical."""
    dim, H, B = 64, 256, 64
    net = make_net(dim)
    cond = synth.make_condition(B, H, seed=1)
    x = synth.make_noise(2, "net.x", B, 4, H)
    t = torch.full((B,), 321, dtype=torch.long)
    with torch.inference_mode():
        y64 = net(x.to(DEV), t.to(DEV), to_dev(cond)).cpu()
        net.hip_engine(DEV).plans.clear()
        torch.cuda.empty_cache()
        y16 = net(x[16:32].to(DEV), t[16:32].to(DEV), to_dev({k: v[16:32] for k, v in cond.items()})).cpu()
    assert torch.isfinite(y64).all() and float(y64.abs().max()) > 0.1
    assert torch.equal(y64[16:32], y16)


def test_reference_shipped_workload_kernel_selection():
    """What `bench.py --config ref48` launches: every 3x3 conv of the d = 48 net at 512 x 512 runs on the F(4x4) kernel (cout 48 / 96 padded to
    its 64-cout tile, reported in DESIGN as the padding loss), the stage resolutions are 512 / 256 / 128 / 64, and no layer falls to F(2x2)."""
    net = make_net(48)
    plan = net.hip_engine(DEV).plan(1, 512, 512)
    convs = [(name, m) for _, _, name, m in plan.step_ops if m and "tiling" in m]
    assert len(convs) == 49
    assert all(name.startswith("nd_conv3x3_wino4") for name, _ in convs), sorted({name for name, _ in convs})
    assert {(m["H"], m["cout"]) for _, m in convs} >= {(512, 48), (256, 96), (128, 192), (64, 384)}


def test_config5_at_512_lsid_and_compose_psnr_match_oracle():
    """BASELINE config 5 at its stated size: LSID.forward on a 512x512x4 frame against the oracle (every conv on conv3x3_wino4 with the
    LeakyReLU prologues -- at 64x64 the narrow layers fall to the F(2x2) kernel), then noise -> clip / compose -> denoise -> PSNR on both sides."""
    from noisediff_amd import io
    from noisediff_amd.spec import lsid_param_spec
    _oracle_threads()
    lsid = make_lsid()
    sd_l = synth.make_state_dict(lsid_param_spec(), 0)
    H = 512
    x = synth.uniform(9, "lsid.x.512", (1, 4, H, H), 0.0, 1.0)
    with torch.inference_mode():
        y = lsid(x.to(DEV)).cpu()
        ref = O.lsid_forward(sd_l, x)
    assert float(ref.abs().max()) > 0.05
    assert close(y.numpy(), ref.numpy(), NET_TOL, atol=5e-5)      # (18 F(4x4) convolutions at 512 x 512 without a norm in between: measured floor 1.9e-5; everywhere else the default 1e-5 holds)
    clean = synth.uniform(9, "lsid.clean.512", (1, 4, H, H), 0.0, 1.0)
    noise = synth.make_noise(9, "lsid.noise.512", 1, 4, H) * 0.1
    with torch.inference_mode():
        den = lsid(io.compose_noisy(noise, clean).to(DEV)).clamp(0, 1).cpu()
    noisy, ref_den, ref_psnr = O.compose_and_denoise(sd_l, noise, clean)
    assert close(den.numpy(), ref_den.numpy(), NET_TOL, atol=5e-5)      # (18 F(4x4) convolutions at 512 x 512 without a norm in between: measured floor 1.9e-5; everywhere else the default 1e-5 holds)
    assert abs(io.psnr(den, clean) - ref_psnr) < 1e-3


def test_sampler_25_step_ddpm_at_256_matches_oracle_along_the_trajectory():
    """The headline configuration (d=64, 256x256x4, pred_v, sigmoid2) through a complete 25-step DDPM chain -- x_T to x_0 with the
    reference's noise draws injected -- compared with the oracle at EVERY step (tools/parity_full_length.py runs all 1000)."""
    _oracle_threads()
    dim, B, H, T = 64, 1, 256, 25
    net = make_net(dim)
    gd = GaussianDiffusion(net, image_size=H, timesteps=T, beta_schedule="sigmoid2").to(DEV)
    cond = synth.make_condition(B, H, seed=1)
    x_T = synth.make_noise(2, "x_T", B, 4, H)
    steps = torch.stack([synth.make_noise(2, f"noise.{i}", B, 4, H) for i in range(T - 1)])
    traj = gd.sample(batch_size=B, condition=to_dev(cond), return_all_timesteps=True, noise={"x_T": x_T, "steps": steps}).cpu()
    ref = O.sample(state_dict(dim), cond, image_size=H, batch_size=B, timesteps=T, x_T=x_T, noise=lambda i, s: steps[i], return_all=True)
    assert traj.shape == ref.shape == (B, T + 1, 4, H, H)
    worst = max(rel_err(traj[:, k].numpy(), ref[:, k].numpy()) for k in range(T + 1))
    assert worst < SAMPLE_TOL / 10, worst
    assert close(traj.numpy(), ref.numpy(), SAMPLE_TOL / 10)              # ... and every element of every x_t within 1e-5 + 1e-3 |ref|                                   # measured ~4e-6 over 1000 steps (profiles/)


def test_config3_batch16_at_256_matches_oracle_on_two_rows():
    """BASELINE config 3 per-GPU shard at its size (d=64, 256x256x4, batch 16): a 3-step DDIM of the whole batch; rows 5 and 12 must equal
    the oracle run on those samples alone (per-sample independence + parity at B=16), rows [6, 8) the same rows run as a shard."""
    _oracle_threads()
    dim, B, H, S = 64, 16, 256, 3
    net = make_net(dim)
    gd = GaussianDiffusion(net, image_size=H, timesteps=1000, sampling_timesteps=S, ddim_sampling_eta=0.5, beta_schedule="sigmoid2").to(DEV)
    cond = synth.make_condition(B, H, seed=1)
    x_T = synth.make_noise(2, "x_T", B, 4, H)
    steps = torch.stack([synth.make_noise(2, f"noise.{i}", B, 4, H) for i in range(S - 1)])
    full = gd.sample(batch_size=B, condition=to_dev(cond), noise={"x_T": x_T, "steps": steps}).cpu()
    for row in (5, 12):                                                  # two rows of the batch of 16 against the oracle run on that sample alone
        one = {k: v[row:row + 1] for k, v in cond.items()}
        ref = O.sample(state_dict(dim), one, image_size=H, batch_size=1, timesteps=1000, sampling_timesteps=S, eta=0.5, x_T=x_T[row:row + 1],
                       noise=lambda i, s, r=row: steps[i][r:r + 1])
        assert close(full[row:row + 1].numpy(), ref.numpy(), SAMPLE_TOL / 10), row
    part = gd.sample(batch_size=2, condition=to_dev({k: v[6:8] for k, v in cond.items()}), noise={"x_T": x_T[6:8], "steps": steps[:, 6:8]}).cpu()
    assert rel_err(part.numpy(), full[6:8].numpy()) < 1e-5


# --------------------------------------------------------------------------- multi-rank product path

def _rank_worker(rank, world, port, q, mode, backend="gloo", total=5):
    """One rank of the sharded HIP sampler: broadcast_weights (the one collective) + sample_sharded, everything on cuda:0
    (the GPU box has one card, so the process group is gloo; on an 8-GPU node the same code runs with backend nccl = RCCL)."""
    import os
    import torch.distributed as dist
    from noisediff_amd.shard import broadcast_weights, sample_sharded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    if backend == "nccl":                                          # RCCL, bound to the device like bench.py --gpus N does
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=DEV)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dim, H, T = 16, 32, 6
    args = SimpleNamespace(dim=dim, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False)
    net = NoiseDiffNet(args)                                       # every rank starts from its own random init ...
    if rank == 0:
        net.load_state_dict(state_dict(dim), strict=True)          # ... only rank 0 holds the checkpoint
    net = net.to(DEV).eval()
    from noisediff_amd.shard import shard_bounds
    lo, hi = shard_bounds(total, rank, world)
    # philox mode: the broadcast carries only the arena slices this job's plans read (recorded before the weights arrive)
    # explicit mode: the network's raw fp32 weights travel and every rank packs its own arena (the default); philox mode: rank 0's
    # packed arena travels, only the slices this job's plans read (recorded before the weights arrive)
    eng = broadcast_weights(net, DEV, src=0, packed=mode == "philox", shapes=[(hi - lo, H, H)] if mode == "philox" else None)
    assert net.hip_engine(DEV) is eng                              # the broadcast weights are the ones the forward uses
    if mode == "philox":
        assert 0 < eng.last_broadcast_bytes < eng.arena.numel() * 4
        if rank != 0:
            absent = next(n for n in eng.slots if n not in eng.valid)
            with pytest.raises(Exception, match="not part of this rank's weight broadcast"):
                eng.p(absent)
    else:
        n_raw = sum(v.numel() for k, v in state_dict(dim).items() if k in eng.slots) * 4
        assert eng.last_broadcast_bytes == n_raw < eng.arena.numel() * 4 and eng.valid is None
    gd = GaussianDiffusion(net, image_size=H, timesteps=T, beta_schedule="sigmoid2").to(DEV)

    def make_cond(lo, hi):
        return to_dev(synth.make_condition(hi - lo, H, seed=1, first_sample=lo, total=total))

    if mode == "philox":
        out = sample_sharded(gd.sample, total, make_cond, seed=21, set_offset=lambda lo: setattr(gd, "sample_offset", lo), gather=True)
    else:
        state = {}

        def fn(batch_size, condition, seed):
            lo = state["lo"]
            noise = {"x_T": synth.make_noise(seed, "x_T", batch_size, 4, H, lo),
                     "steps": torch.stack([synth.make_noise(seed, f"noise.{i}", batch_size, 4, H, lo) for i in range(T - 1)])}
            return gd.sample(batch_size=batch_size, condition=condition, noise=noise)

        out = sample_sharded(fn, total, make_cond, seed=2, set_offset=lambda lo: state.update(lo=lo), gather=True)
    if backend == "nccl":          # a world of one skips sample_sharded's gather: run the collectives of the N > 1 path on device tensors anyway
        parts = [torch.empty_like(out) for _ in range(world)]
        dist.all_gather(parts, out.contiguous())
        cs = out.double().abs().sum().reshape(1)
        lo_, hi_ = cs.clone(), cs.clone()
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        assert torch.equal(parts[rank], out) and lo_.item() == hi_.item() == cs.item()
        assert dist.get_backend() == "nccl" and out.is_cuda
    if rank == 0:
        q.put(out.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["explicit", "philox"])
def test_rccl_process_group_of_one_rank_runs_the_collectives_of_the_sharded_sampler(mode):
    """The GPU box has one card and RCCL refuses two ranks on one device, so the N > 1 tests below run on gloo.  This one runs the SAME
    worker on backend nccl (= RCCL) with a world of one rank: communicator creation bound to the device (``device_id``), the weight
    broadcast (raw weights / packed plan slices), the barrier and the all-gather of the patches all execute inside librccl, with device
    tensors -- what bench.py --gpus N and shard.py call on an 8-GPU node, minus the peer transport."""
    import os
    import torch.multiprocessing as mp
    dim, H, T, total = 16, 32, 6, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31600 + (os.getpid() + (0 if mode == "explicit" else 1)) % 2000
    p = ctx.Process(target=_rank_worker, args=(0, 1, port, q, mode, "nccl"))
    p.start()
    got = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    net = make_net(dim)
    gd = GaussianDiffusion(net, image_size=H, timesteps=T, beta_schedule="sigmoid2").to(DEV)
    cond = synth.make_condition(total, H, seed=1)
    if mode == "philox":
        one = gd.sample(batch_size=total, condition=to_dev(cond), seed=21).cpu().numpy()
    else:
        x_T = synth.make_noise(2, "x_T", total, 4, H)
        steps = torch.stack([synth.make_noise(2, f"noise.{i}", total, 4, H) for i in range(T - 1)])
        one = gd.sample(batch_size=total, condition=to_dev(cond), noise={"x_T": x_T, "steps": steps}).cpu().numpy()
    assert np.isfinite(got).all() and rel_err(got, one) < 1e-5


@pytest.mark.parametrize("world,total", [(2, 5), (5, 19)])
@pytest.mark.parametrize("mode", ["explicit", "philox"])
def test_n_rank_hip_sampler_equals_single_rank(mode, world, total):
    """2 ranks (ragged 3 + 2 split of 5 patches) and 5 ranks (19 patches: 4 + 4 + 4 + 4 + 3 -- the GPU box admits six processes on its card, this one included;
    the 8-rank case of SURVEY section 4 runs on the CPU path, tests/test_host.py): Engine.broadcast -> adopt_engine -> sharded HIP sampling -> all-gather
    == the same patches sampled by one process; explicit-noise mode is also checked against the oracle."""
    import os
    import torch.multiprocessing as mp
    dim, H, T = 16, 32, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + (0 if mode == "explicit" else 1) + 2 * world) % 2000
    procs = [ctx.Process(target=_rank_worker, args=(r, world, port, q, mode, "gloo", total)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    net = make_net(dim)
    gd = GaussianDiffusion(net, image_size=H, timesteps=T, beta_schedule="sigmoid2").to(DEV)
    cond = synth.make_condition(total, H, seed=1)
    if mode == "philox":
        one = gd.sample(batch_size=total, condition=to_dev(cond), seed=21).cpu().numpy()
        assert np.isfinite(got).all() and rel_err(got, one) < 1e-5
        return
    x_T = synth.make_noise(2, "x_T", total, 4, H)
    steps = torch.stack([synth.make_noise(2, f"noise.{i}", total, 4, H) for i in range(T - 1)])
    one = gd.sample(batch_size=total, condition=to_dev(cond), noise={"x_T": x_T, "steps": steps}).cpu().numpy()
    assert rel_err(got, one) < 1e-5
    ref = O.sample(state_dict(dim), cond, image_size=H, batch_size=total, timesteps=T, x_T=x_T, noise=lambda i, s: steps[i])
    assert rel_err(got, ref.numpy()) < SAMPLE_TOL / 10


def test_bench_self_launcher_two_ranks_on_one_device():
    """`python bench.py --gpus 2` as the driver starts it: a fresh child process that spawns the two rank processes itself (both on cuda:0 here, gloo
    rendezvous on 127.0.0.1), one weight broadcast, zero per-step collectives, one JSON line from rank 0; and a job size that contradicts WORLD_SIZE is
    refused with exit code 2 instead of being reported as something it is not (reference entry: models/modules.py:73-83)."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--one-device", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--soak-s", "0",
           "--no-cpu", "--no-roofline", "--size", "64", "--batch", "2", "--timesteps", "8"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    mg = line["multi_gpu"]
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 4 and line["scaling"] == "weak"
    assert mg["ranks_in_broadcast"] == 2 and mg["arena_checksum_equal_on_all_ranks"] is True and mg["per_step_collectives"] == 0
    assert len(mg["ms_per_step_by_rank_wall"]) == 2 and len(mg["ms_per_step_by_rank_gpu_events"]) == 2
    assert "bench.py started the ranks itself" in mg["launcher"]
    bad = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "3", "--one-device", "--backend", "gloo", "--steps", "1", "--warmup", "0",
                          "--soak-s", "0", "--no-cpu", "--no-roofline"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True,
                         timeout=300)
    assert bad.returncode == 2 and "WORLD_SIZE=2" in bad.stderr


def test_split_chains_of_a_batch_cut_into_pieces_equal_the_whole_batch_bit_for_bit(monkeypatch):
    """nd_pointwise_chain_split_nhwc_f32 addresses its tensors through 32-bit offsets; Plan.chain runs a batch beyond that as equal pieces of whole samples
    (B >= 256 at 256 x 256, d = 64).  Forced here with a small limit: the forward of a batch of 5 in pieces of 2 + 2 + 1 equals the one-launch forward bit for bit."""
    from noisediff_amd import engine as E
    dim, H, B = 16, 32, 5
    sd = state_dict(dim)
    cond = to_dev(synth.make_condition(B, H, seed=1))
    x = synth.make_noise(3, "piece.x", B, 4, H).to(DEV)
    t = torch.tensor([3, 500, 999, 0, 77], dtype=torch.long, device=DEV)
    outs = []
    for limit in (E._CHAIN_SPLIT_LIMIT, 2 * 4 * H * H * dim):           # the second: two samples of a dim-channel tensor per launch
        monkeypatch.setattr(E, "_CHAIN_SPLIT_LIMIT", limit)
        net = make_net(dim)
        plan = net.hip_engine(DEV).plan(B, H, H)
        n_chain = sum(1 for op in plan.step_ops if op[2] == "nd_pointwise_chain_split_nhwc_f32")
        plan.set_condition(cond)
        outs.append((plan.forward(x, t).clone(), n_chain))
    assert outs[1][1] > outs[0][1] > 0                                   # more launches, same layers
    assert torch.equal(outs[0][0], outs[1][0])
