"""SURVEY 8f-3, second half: upstream's per-stage attention wiring -- LinearAttention / Attention behind every stage's second ResnetBlock,
the classes the reference defines (models/archs/Diffusion_arch.py:198-266), computes the flags for (:467-468,509-518) and drops.

CPU part: spec and oracle against the fixture captured from the real reference network with forward hooks (tests/golden/capture_stage_attn.py).
GPU part (-m gpu): ``NoiseDiffNet(args)`` with ``args.stage_attn = True`` on the HIP kernels."""
from types import SimpleNamespace

import pytest
import torch

from noisediff_amd import synth
from noisediff_amd.spec import STAGE_ATTN_REFERENCE, noisediff_param_spec, normalize_stage_attn, stage_attention_param_spec
from oracle import noisediff_oracle as O
from util import noise_fn, rel_err, sub

DIM, B, S = 16, 2, 64
KINDS = STAGE_ATTN_REFERENCE
TAPS = ("down0", "down3", "mid", "up0", "up3")


def _inputs(dim=DIM, size=S):
    sd = synth.make_state_dict(noisediff_param_spec(dim), 0)
    sd.update(synth.make_state_dict(stage_attention_param_spec(dim, KINDS), 0))
    return sd, synth.make_condition(B, size, seed=1), synth.make_noise(4, "sa.x", B, 4, size)


def test_stage_attention_spec():
    assert normalize_stage_attn(None) is None and normalize_stage_attn(False) is None and normalize_stage_attn((None,) * 4) is None
    assert normalize_stage_attn(True) == KINDS == ("linear", "linear", "linear", "full")       # the reference's full_attn = (False, False, False, True)
    assert normalize_stage_attn((False, False, False, True)) == KINDS                          # upstream's booleans
    assert normalize_stage_attn(("full", None, None, "linear")) == ("full", None, None, "linear")
    for bad in ((True,) * 3, ("linear", "x", None, None), "linear"):
        with pytest.raises(ValueError):
            normalize_stage_attn(bad)
    spec = stage_attention_param_spec(DIM, KINDS)
    names = {p.name: tuple(p.shape) for p in spec}
    assert len(spec) == 2 * (3 * 5 + 4)                                                        # names / shapes were loaded strict=True into the reference classes at capture time
    assert names["down_attns.0.to_out.1.g"] == (1, DIM, 1, 1) and names["down_attns.3.to_out.weight"] == (4 * DIM, 128, 1, 1)
    assert names["up_attns.0.to_qkv.weight"] == (384, 8 * DIM, 1, 1) and names["up_attns.3.to_qkv.weight"] == (384, DIM, 1, 1)
    assert len(stage_attention_param_spec(DIM, ("full", None, None, None))) == 8


def test_stage_attention_oracle_matches_the_reference(golden):
    sd, cond, x = _inputs()
    with torch.no_grad():
        for t in (3, 500, 999):
            taps = {}
            y = O.noisediff_forward(sd, x, torch.full((B,), t, dtype=torch.long), cond, taps=taps, stage_attention=KINDS)
            assert rel_err(y.numpy(), golden("stage_attn", f"sa.fwd.t{t}")) < 2e-5, t
            if t == 500:
                for k in TAPS:
                    assert rel_err(sub(taps[k]), golden("stage_attn", f"sa.tap.{k}")) < 2e-5, k
        plain = O.noisediff_forward(sd, x, torch.full((B,), 500, dtype=torch.long), cond)
        assert rel_err(plain.numpy(), golden("stage_attn", "sa.fwd.t500")) > 1e-3              # the wiring changes the function
        res = O.sample(sd, cond, image_size=S, batch_size=B, timesteps=1000, sampling_timesteps=6, eta=0.5,
                       x_T=synth.make_noise(2, "x_T", B, 4, S), noise=noise_fn(2, B, 4, S), stage_attention=KINDS)
    assert rel_err(res.numpy(), golden("stage_attn", "sa.samp.ddim6")) < 1e-4


# --------------------------------------------------------------------------------------------- HIP (MI355X)

def _hip_net(dim, kinds=True):
    import noisediff_amd as nd
    net = nd.NoiseDiffNet(SimpleNamespace(dim=dim, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False, stage_attn=kinds))
    return net


@pytest.mark.gpu
def test_stage_attention_hip_forward_and_sampler(golden):
    import noisediff_amd as nd
    dev = torch.device("cuda", 0)
    sd, cond, x = _inputs()
    net = _hip_net(DIM)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    cond_d = {k: v.to(dev) for k, v in cond.items()}
    with torch.inference_mode():
        for t in (3, 500, 999):
            y = net(x.to(dev), torch.full((B,), t, dtype=torch.long, device=dev), cond_d)
            assert rel_err(y.cpu().numpy(), golden("stage_attn", f"sa.fwd.t{t}")) < 2e-4, t
        plan = net.hip_engine(dev).plan(B, S, S, debug=True)
        plan.set_condition(cond_d)
        plan.forward(x.to(dev), torch.full((B,), 500, dtype=torch.long, device=dev))
        for k in TAPS:
            got = plan.taps[k].permute(0, 3, 1, 2).contiguous().cpu()
            assert rel_err(sub(got), golden("stage_attn", f"sa.tap.{k}")) < 2e-4, k
        gd = nd.GaussianDiffusion(torch.nn.DataParallel(net), image_size=S, timesteps=1000, sampling_timesteps=6, ddim_sampling_eta=0.5,
                                  beta_schedule="sigmoid2", objective="pred_v").to(dev)
        steps = torch.stack([synth.make_noise(2, f"noise.{i}", B, 4, S) for i in range(5)])
        res = gd.sample(batch_size=B, condition=cond_d, noise={"x_T": synth.make_noise(2, "x_T", B, 4, S), "steps": steps})
    assert rel_err(res.cpu().numpy(), golden("stage_attn", "sa.samp.ddim6")) < 1e-3             # north-star tolerance
    # under autograd the same module runs the differentiable graph (r5; loss and gradients against the reference: tests/test_trainable.py) -- the same function
    y = net(x.to(dev), torch.full((B,), 500, dtype=torch.long, device=dev), cond_d)
    assert y.requires_grad and rel_err(y.detach().cpu().numpy(), golden("stage_attn", "sa.fwd.t500")) < 2e-4


@pytest.mark.gpu
def test_stage_attention_hip_at_bench_width_matches_oracle():
    """d=64 at 128x128 (LinearAttention over 16384 / 4096 / 1024 tokens, full Attention over 256; Winograd convs, chains, wide pointwise
    kernels all active), a mixed wiring with a stage left out, against the oracle (itself pinned by the d=16 reference golden above)."""
    dev = torch.device("cuda", 0)
    dim, size, kinds = 64, 128, ("linear", None, "linear", "full")
    sd = synth.make_state_dict(noisediff_param_spec(dim), 0)
    sd.update(synth.make_state_dict(stage_attention_param_spec(dim, kinds), 0))
    cond = synth.make_condition(B, size, seed=1)
    x = synth.make_noise(4, "sa64.x", B, 4, size)
    t = torch.tensor([17, 803], dtype=torch.long)
    net = _hip_net(dim, kinds)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    with torch.inference_mode():
        y = net(x.to(dev), t.to(dev), {k: v.to(dev) for k, v in cond.items()}).cpu()
    torch.set_num_threads(16)
    with torch.no_grad():
        ref = O.noisediff_forward(sd, x, t, cond, stage_attention=kinds)
    assert rel_err(y.numpy(), ref.numpy()) < 2e-4
