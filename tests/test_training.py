"""Training entry points of the diffusion wrapper (GaussianDiffusion.forward -> loss), CPU.

`q_sample` / `p_losses` / `forward` are plain differentiable PyTorch around ANY nn.Module with the arch plug-in
signature.  The fixture (tests/golden/training.npz, captured from the reference by capture_training.py) pins x_t, the
pred_v target, the loss of all three objectives, the offset-noise branch, `forward` and parameter gradients.  The model
used here is the CPU oracle's functional forward wrapped as an nn.Module (test infrastructure, differentiable).
"""
import copy

import numpy as np
import pytest
import torch
from torch import nn

from noisediff_amd import synth
from noisediff_amd.diffusion import GaussianDiffusion
from oracle import noisediff_oracle as O
from util import state_dict, sub

DIM, B, H, T = 16, 2, 32, 1000
GRAD_KEYS = ["final_conv.weight", "downs.0.0.block1.proj.weight", "time_mlp.1.weight", "mid_block1.block2.norm.weight",
             "ups.3.2.ff.net.2.weight", "shot_mlp1.fc1.weight", "pos_block1.mlp.1.bias", "iso_embed.weight"]


class OracleNet(nn.Module):
    """The oracle forward as a differentiable module with the reference's plug-in surface (Diffusion_arch.py:447-646)."""
    channels = out_dim = 4
    self_condition = False
    random_or_learned_sinusoidal_cond = False

    def __init__(self, sd):
        super().__init__()
        self.names = list(sd)
        self.values = nn.ParameterList([nn.Parameter(v.clone()) for v in sd.values()])

    def forward(self, x, time, condition):
        return O.noisediff_forward(dict(zip(self.names, self.values)), x, time, condition)


def _inputs():
    return (synth.uniform(5, "train.x0", (B, 4, H, H), -1.0, 1.0), synth.make_noise(5, "train.noise", B, 4, H),
            torch.tensor([3, 777], dtype=torch.long), synth.make_condition(B, H, seed=1))


def _gd(objective="pred_v", **kw):
    net = OracleNet(state_dict(DIM))
    return net, GaussianDiffusion(nn.DataParallel(net), image_size=H, timesteps=T, beta_schedule="sigmoid2", objective=objective, **kw)


def test_q_sample_and_v_target_match_the_reference(golden):
    x0, noise, t, _ = _inputs()
    _, gd = _gd()
    assert np.array_equal(gd.q_sample(x0, t, noise).numpy(), golden("training", "train.x_t"))
    assert np.array_equal(gd.predict_v(x0, t, noise).numpy(), golden("training", "train.v_target"))
    torch.manual_seed(3)
    a = gd.q_sample(x0, t)
    torch.manual_seed(3)
    assert torch.equal(a, gd.q_sample(x0, t, torch.randn_like(x0)))          # default noise = torch.randn_like (:474)


@pytest.mark.parametrize("objective", ["pred_v", "pred_noise", "pred_x0"])
def test_p_losses_match_the_reference(golden, objective):
    x0, noise, t, cond = _inputs()
    _, gd = _gd(objective)
    loss = gd.p_losses(x0, t, cond, noise=noise.clone())
    ref = float(golden("training", f"train.loss.{objective}"))
    assert loss.ndim == 0 and float(loss) == pytest.approx(ref, rel=2e-5)


def test_gradients_offset_noise_and_forward_match_the_reference(golden, monkeypatch):
    x0, noise, t, cond = _inputs()
    net, gd = _gd()
    gd.p_losses(x0, t, cond, noise=noise.clone()).backward()
    grads = dict(zip(net.names, (p.grad for p in net.values)))
    for k in GRAD_KEYS:
        ref = golden("training", f"train.grad.{k}")
        got = sub(grads[k], 2048)
        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), k
    # dead parameters of the 1-token cross attention get an exactly-zero / absent gradient in the reference too
    sq = sum(float((g.double() ** 2).sum()) for g in grads.values() if g is not None)
    assert sq == pytest.approx(float(golden("training", "train.grad_sq_norm")), rel=1e-4)
    # offset noise (one draw per (sample, channel)) and forward() with the reference's draws patched in
    monkeypatch.setattr(torch, "randn", lambda shape, *a, **k: synth.uniform(5, "train.offset", tuple(shape), -1.0, 1.0))
    monkeypatch.setattr(torch, "randn_like", lambda t_, *a, **k: synth.make_noise(5, "train.noise", B, 4, H))
    monkeypatch.setattr(torch, "randint", lambda lo, hi, shape, *a, **k: torch.tensor([3, 777], dtype=torch.long))
    keep = noise.clone()
    loss = gd.p_losses(x0, t, cond, noise=noise, offset_noise_strength=0.1)
    assert float(loss) == pytest.approx(float(golden("training", "train.loss.pred_v.offset0.1")), rel=2e-5)
    assert torch.equal(noise, keep)                                           # the caller's tensor is not modified
    assert float(gd(x0, cond)) == pytest.approx(float(golden("training", "train.loss.pred_v.forward")), rel=2e-5)
    with pytest.raises(AssertionError, match="height and width of image must be 32"):
        gd(torch.zeros(1, 4, 16, 16), cond)


def test_wrapper_copies_and_sampling_still_needs_the_hip_net():
    net, gd = _gd()
    with pytest.raises(TypeError, match="HIP engine"):
        gd.sample(batch_size=1, condition=synth.make_condition(1, H, seed=1))
    gd._loop_cache["x"] = object()                # device loops (ctypes handles) never travel with a copy
    assert copy.deepcopy(gd)._loop_cache == {}
    gd._loop_cache = {}
    auto = GaussianDiffusion(net, image_size=H, timesteps=T, beta_schedule="sigmoid2", auto_normalize=True)
    assert torch.equal(auto.normalize(torch.tensor([0.0, 1.0])), torch.tensor([-1.0, 1.0]))
