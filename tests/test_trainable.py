"""noisediff_amd.TrainableNoiseDiffNet: the differentiable network of the training path.

CPU: its parameter names are the reference's, its forward equals the oracle's (which the reference's golden activations pin,
tests/test_oracle.py) and -- wrapped by GaussianDiffusion -- it reproduces the reference's training loss and parameter gradients
(tests/golden/training.npz, captured from the reference by tests/golden/capture_training.py).
GPU: with .hip() (3x3 convolutions and GroupNorms on the HIP library, forward and backward) loss and gradients stay the same, and the
trained weights load into the HIP sampler unchanged."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from torch import nn

from noisediff_amd import GaussianDiffusion, TrainableNoiseDiffNet, synth
from noisediff_amd.spec import noisediff_param_spec
from oracle import noisediff_oracle as O
from util import rel_err, state_dict, sub

DIM, B, H, T = 16, 2, 32, 1000


def golden_value(key):
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "training.npz"))[key]
GRAD_KEYS = ["final_conv.weight", "downs.0.0.block1.proj.weight", "time_mlp.1.weight", "mid_block1.block2.norm.weight",
             "ups.3.2.ff.net.2.weight", "shot_mlp1.fc1.weight", "pos_block1.mlp.1.bias", "iso_embed.weight"]


def _net(dim=DIM):
    net = TrainableNoiseDiffNet(SimpleNamespace(dim=dim, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False, phase="train"))
    net.load_state_dict(state_dict(dim), strict=True)
    return net


def _inputs():
    return (synth.uniform(5, "train.x0", (B, 4, H, H), -1.0, 1.0), synth.make_noise(5, "train.noise", B, 4, H),
            torch.tensor([3, 777], dtype=torch.long), synth.make_condition(B, H, seed=1))


def test_parameter_names_and_default_init_follow_the_spec():
    net = TrainableNoiseDiffNet(SimpleNamespace(dim=DIM))
    spec = noisediff_param_spec(DIM)
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == [(p.name, tuple(p.shape)) for p in spec]
    assert all(v.requires_grad for v in net.parameters())
    with pytest.raises(ValueError):
        TrainableNoiseDiffNet(SimpleNamespace(dim=DIM, self_condition=True))


def test_forward_equals_the_oracle():
    net = _net()
    x0, _, t, cond = _inputs()
    with torch.no_grad():
        got = net(x0, t, cond)
        ref = O.noisediff_forward(state_dict(DIM), x0, t, cond)
    assert got.shape == ref.shape == (B, 4, H, H)
    assert rel_err(got.numpy(), ref.numpy()) < 1e-6


def test_loss_and_gradients_match_the_reference(golden):
    x0, noise, t, cond = _inputs()
    net = _net()
    gd = GaussianDiffusion(nn.DataParallel(net), image_size=H, timesteps=T, beta_schedule="sigmoid2", objective="pred_v")
    loss = gd.p_losses(x0, t, cond, noise=noise.clone())
    assert float(loss.detach()) == pytest.approx(float(golden("training", "train.loss.pred_v")), rel=2e-5)
    loss.backward()
    grads = {k: p.grad for k, p in net.named_parameters()}
    for k in GRAD_KEYS:
        ref = golden("training", f"train.grad.{k}")
        got = sub(grads[k], 2048)
        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), k
    sq = sum(float((g.double() ** 2).sum()) for g in grads.values() if g is not None)
    assert sq == pytest.approx(float(golden("training", "train.grad_sq_norm")), rel=1e-4)
    # every parameter gets a gradient (DDP needs that); the queries / keys of the one-token cross attention and the LayerNorm in front
    # of them are dead in the reference too: exactly zero
    assert all(g is not None for g in grads.values())
    dead = [k for k in grads if k.endswith((".attn.to_q.weight", ".attn.to_k.weight", ".norm1.weight", ".norm1.bias"))]
    assert len(dead) == 9 * 4 and all(float(grads[k].abs().max()) == 0.0 for k in dead)


ARCH_CASES = {      # case of tests/golden/training_archs.npz -> (args of the network, gradients the fixture holds)
    "UNet_PosEmbV2": (dict(arch="UNet_PosEmbV2"), ["final_conv.weight", "cond_init_conv.weight", "cond_concat_conv.weight", "downs.0.0.block1.proj.weight",
                                                    "time_mlp.1.weight", "mid_block1.block2.norm.weight", "pos_block1.block1.proj.weight"]),
    "UNet_PosEmbV2_NoPosition": (dict(arch="UNet_PosEmbV2_NoPosition"), ["final_conv.weight", "cond_init_conv.weight", "cond_concat_conv.weight",
                                                                          "downs.0.0.block1.proj.weight", "time_mlp.1.weight", "pos_block1.block1.proj.weight"]),
    "UNet_PosEmbV2_CameraCond": (dict(arch="UNet_PosEmbV2_CameraCond"), ["final_conv.weight", "cond_concat_conv.weight", "downs.1.2.ff.net.2.weight", "iso_embed.weight",
                                                                          "time_mlp.1.weight"]),
    "mid_attn": (dict(mid_attn=True), ["final_conv.weight", "downs.3.1.block2.proj.weight", "mid_block2.block1.proj.weight", "mid_attn.to_qkv.weight",
                                       "mid_attn.to_out.weight", "mid_attn.norm.g"]),
    "stage_attn": (dict(stage_attn=True), ["final_conv.weight", "downs.3.1.block2.proj.weight", "down_attns.0.to_qkv.weight", "down_attns.3.to_out.weight",
                                           "up_attns.0.to_qkv.weight", "up_attns.3.to_out.1.g"]),
}


def _arch_net(case, cls=TrainableNoiseDiffNet):
    kw, _ = ARCH_CASES[case]
    net = cls(SimpleNamespace(dim=DIM, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False, phase="train", **kw))
    net.load_state_dict(_arch_state_dict(net), strict=True)
    return net


def _arch_state_dict(net):
    """The synthetic weights of the capture script: every tensor from its own hash stream (seed 0), keyed by its state-dict name."""
    from noisediff_amd.spec import arch_param_spec, attention_param_spec, stage_attention_param_spec
    spec = list(arch_param_spec(net.arch if hasattr(net, "arch") else net.ARCH, DIM, 4))
    if net.has_mid_attn:
        spec += attention_param_spec("mid_attn", 8 * DIM)
    if net.stage_attn:
        spec += stage_attention_param_spec(DIM, net.stage_attn)
    return synth.make_state_dict(spec, 0)


def _check_arch_loss_and_gradients(golden, case, net, loss, loss_rel, grad_tol, sq_rel):
    assert float(loss.detach()) == pytest.approx(float(golden("training_archs", f"{case}.loss")), rel=loss_rel)
    grads = {k: p.grad for k, p in net.named_parameters()}
    assert sum(1 for g in grads.values() if g is not None) == int(golden("training_archs", f"{case}.n_params_with_grad"))
    for k in ARCH_CASES[case][1]:
        ref = golden("training_archs", f"{case}.grad.{k}")
        got = sub(grads[k].cpu(), 2048)
        assert np.abs(got - ref).max() <= grad_tol * max(1.0, np.abs(ref).max()), (case, k)
    sq = sum(float((g.double() ** 2).sum()) for g in grads.values() if g is not None)
    assert sq == pytest.approx(float(golden("training_archs", f"{case}.grad_sq_norm")), rel=sq_rel)


@pytest.mark.parametrize("case", sorted(ARCH_CASES))
def test_other_architectures_loss_and_gradients_match_the_reference(golden, case):
    """The ``UNet_PosEmbV2*`` ablation nets (others_arch.py:364-985), NoiseDiffNet with the mid-block Attention of BASELINE config 4 and NoiseDiffNet with
    upstream's per-stage LinearAttention / Attention wiring under ``p_losses`` (denoising_diffusion_pytorch.py:481-531): loss, the squared norm of the
    whole gradient, the number of parameters that receive one and samples of parameter gradients equal the reference's (tests/golden/training_archs.npz,
    captured from the reference's own classes by tests/golden/capture_training_archs.py)."""
    x0, noise, t, cond = _inputs()
    net = _arch_net(case)
    gd = GaussianDiffusion(nn.DataParallel(net), image_size=H, timesteps=T, beta_schedule="sigmoid2", objective="pred_v")
    ref_cond = cond["clean_img"] if case == "UNet_PosEmbV2_NoPosition" else cond      # others_arch.py:658
    loss = gd.p_losses(x0, t, ref_cond, noise=noise.clone())
    loss.backward()
    _check_arch_loss_and_gradients(golden, case, net, loss, 2e-5, 2e-5, 1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(ARCH_CASES))
def test_the_drop_in_networks_of_every_architecture_train_under_autograd(golden, case):
    """``noisediff_amd.UNet_PosEmbV2*`` and ``noisediff_amd.NoiseDiffNet`` with ``mid_attn`` / ``stage_attn`` under autograd on the differentiable HIP operators
    (the modules the registry hands to define_G: they sample on the fused engine and train through the same ``forward``): loss and gradients against the
    reference's, and the differentiable forward equals the fused engine's."""
    import noisediff_amd
    dev = torch.device("cuda", 0)
    x0, noise, t, cond = _inputs()
    x0, noise, t = x0.to(dev), noise.to(dev), t.to(dev)
    cond_dev = {k: (v if k == "iso_ratio_idx" else v.to(dev)) for k, v in cond.items()}
    kw, _ = ARCH_CASES[case]
    cls = getattr(noisediff_amd, kw.get("arch", "NoiseDiffNet"))
    net = cls(SimpleNamespace(dim=DIM, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False, phase="train",
                              **{k: v for k, v in kw.items() if k != "arch"}))
    net.load_state_dict(_arch_state_dict(net), strict=True)
    net = net.to(dev).train()
    ref_cond = cond_dev["clean_img"] if case == "UNet_PosEmbV2_NoPosition" else cond_dev
    gd = GaussianDiffusion(nn.DataParallel(net, device_ids=[0]), image_size=H, timesteps=T, beta_schedule="sigmoid2", objective="pred_v").to(dev)
    with torch.no_grad():
        fused = net(x0, t, ref_cond).clone()                                   # the sampling engine
    loss = gd.p_losses(x0, t, ref_cond, noise=noise.clone())
    loss.backward()
    _check_arch_loss_and_gradients(golden, case, net, loss, 5e-5, 2e-4, 1e-3)
    with torch.enable_grad():
        y = net(x0, t, ref_cond)
    assert y.requires_grad and rel_err(y.detach().cpu().numpy(), fused.cpu().numpy()) < 2e-4


def test_cross_attention_general_form_equals_the_one_token_identity():
    from noisediff_amd.trainable import _Ops
    net = _net()
    o = _Ops(dict(net.named_parameters()), False)
    x = synth.uniform(9, "ca.x", (2, 50, DIM), -1.0, 1.0)
    ctx = synth.uniform(9, "ca.ctx", (2, 1, 16), -1.0, 1.0)
    with torch.no_grad():
        fast = o.cross_attention("shot_attn.attn", x, ctx).expand(2, 50, DIM)
        two = o.cross_attention("shot_attn.attn", x, torch.cat((ctx, ctx), dim=1))           # two identical keys: the general path, same result
    assert rel_err(fast.numpy(), two.numpy()) < 1e-6


@pytest.mark.gpu
def test_hip_convs_and_norms_keep_loss_and_gradients_and_weights_load_into_the_sampler(golden):
    from noisediff_amd import NoiseDiffNet
    dev = torch.device("cuda", 0)
    x0, noise, t, cond = _inputs()
    x0, noise, t = x0.to(dev), noise.to(dev), t.to(dev)
    cond = {k: v.to(dev) for k, v in cond.items()}
    out = []
    for hip in (False, True):
        net = _net().to(dev).hip(hip)
        gd = GaussianDiffusion(net, image_size=H, timesteps=T, beta_schedule="sigmoid2", objective="pred_v").to(dev)
        loss = gd.p_losses(x0, t, cond, noise=noise.clone())
        loss.backward()
        out.append((float(loss.detach()), {k: p.grad.detach().cpu() for k, p in net.named_parameters() if p.grad is not None}))
    assert out[1][0] == pytest.approx(out[0][0], rel=2e-5)
    assert out[0][1].keys() == out[1][1].keys()
    # ... and both paths ON THE GPU reproduce the REFERENCE's loss and parameter gradients (tests/golden/training.npz, captured from
    # models/denoising_diffusion_pytorch.py:481-531 driving the reference net): the .hip() kernels are pinned to the reference itself,
    # not only to PyTorch-on-GPU
    for loss_val, grads in out:
        assert loss_val == pytest.approx(float(golden("training", "train.loss.pred_v")), rel=5e-5)
        for k in GRAD_KEYS:
            ref = golden("training", f"train.grad.{k}")
            got = sub(grads[k], 2048)
            assert np.abs(got - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), k
        sq = sum(float((g.double() ** 2).sum()) for g in grads.values())
        assert sq == pytest.approx(float(golden("training", "train.grad_sq_norm")), rel=1e-3)
    for k in out[0][1]:
        assert rel_err(out[1][1][k].numpy(), out[0][1][k].numpy()) < 5e-4, k
    # nothing of a .hip() network is left on PyTorch's own kernels since r5 (the 7x7 stem, pos_enc's 2 -> 8 convolution and LayerNorms of any width % 4 included);
    # whatever would be is said, not silent: `trainable.FALLBACKS`
    from noisediff_amd import trainable
    assert trainable.FALLBACKS == {}, trainable.FALLBACKS
    # a step of Adam on the accelerated net, then its weights sample on the HIP network
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    opt.step()
    hipnet = NoiseDiffNet(SimpleNamespace(dim=DIM, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False))
    hipnet.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()}, strict=True)
    hipnet = hipnet.to(dev).eval()
    with torch.no_grad():
        ref = net.hip(False)(x0, t, cond)
        y = hipnet(x0, t, cond)
    assert rel_err(y.cpu().numpy(), ref.cpu().numpy()) < 2e-4


@pytest.mark.gpu
def test_the_drop_in_network_trains_under_autograd(golden):
    """``noisediff_amd.NoiseDiffNet`` itself under ``p_losses`` (models/denoising_diffusion_pytorch.py:481-531 through the nn.DataParallel
    wrapper of models/modules.py:81): loss and parameter gradients equal the reference's (tests/golden/training.npz); after an optimizer
    step the SAME module samples on the fused engine with the updated weights (the engine repacks when parameter versions change)."""
    from noisediff_amd import NoiseDiffNet
    dev = torch.device("cuda", 0)
    x0, noise, t, cond = _inputs()
    x0, noise, t = x0.to(dev), noise.to(dev), t.to(dev)
    cond_dev = {k: (v if k == "iso_ratio_idx" else v.to(dev)) for k, v in cond.items()}       # the trainer keeps iso_ratio_idx on the CPU
    net = NoiseDiffNet(SimpleNamespace(dim=DIM, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False, phase="train"))
    net.load_state_dict(state_dict(DIM), strict=True)
    net = net.to(dev).train()
    gd = GaussianDiffusion(nn.DataParallel(net, device_ids=[0]), image_size=H, timesteps=T, beta_schedule="sigmoid2", objective="pred_v").to(dev)
    with torch.no_grad():
        before = net(x0, t, cond_dev).clone()                                  # fused engine
    loss = gd.p_losses(x0, t, cond_dev, noise=noise.clone())
    assert float(loss.detach()) == pytest.approx(float(golden("training", "train.loss.pred_v")), rel=5e-5)
    loss.backward()
    grads = {k: p.grad for k, p in net.named_parameters()}
    assert all(g is not None for g in grads.values())
    for k in GRAD_KEYS:
        ref = golden("training", f"train.grad.{k}")
        assert np.abs(sub(grads[k].cpu(), 2048) - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), k
    sq = sum(float((g.double() ** 2).sum()) for g in grads.values())
    assert sq == pytest.approx(float(golden("training", "train.grad_sq_norm")), rel=1e-3)
    # the differentiable forward and the fused engine are the same function of the same parameters
    with torch.enable_grad():
        y_train = net(x0, t, cond_dev)
    assert y_train.requires_grad and rel_err(y_train.detach().cpu().numpy(), before.cpu().numpy()) < 2e-4
    torch.optim.Adam(net.parameters(), lr=1e-3).step()
    with torch.no_grad():
        after = net(x0, t, cond_dev)                                           # engine repacked from the updated parameters
        ref_after = net._forward_autograd(x0, t, cond_dev)
    assert not torch.equal(after, before)
    assert rel_err(after.cpu().numpy(), ref_after.cpu().numpy()) < 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [64, 48])
def test_no_layer_of_the_hip_network_falls_back_to_pytorch(dim):
    """d = 64 and d = 48 (the reference's shipped width, script.sh:10) with .hip(): every convolution (3x3, 1x1, the 7x7 stem, pos_enc's 2 -> 8), Linear,
    GroupNorm and LayerNorm (C = 48 k included) of a training step runs on the HIP library -- `trainable.FALLBACKS` stays empty -- and the step's loss and
    gradients equal the plain PyTorch evaluation of the same graph."""
    from noisediff_amd import trainable
    dev = torch.device("cuda", 0)
    S = 64
    x0, noise = synth.uniform(5, "fb.x0", (2, 4, S, S), -1.0, 1.0).to(dev), synth.make_noise(5, "fb.noise", 2, 4, S).to(dev)
    t = torch.tensor([3, 777], dtype=torch.long, device=dev)
    cond = {k: v.to(dev) for k, v in synth.make_condition(2, S, seed=1).items()}
    res = []
    for hip in (False, True):
        net = TrainableNoiseDiffNet(SimpleNamespace(dim=dim)).to(dev)
        net.load_state_dict(state_dict(dim), strict=True)
        if hip:
            trainable.FALLBACKS.clear()
            net.hip()
        gd = GaussianDiffusion(net, image_size=S, timesteps=T, beta_schedule="sigmoid2", objective="pred_v").to(dev)
        loss = gd.p_losses(x0, t, cond, noise=noise.clone())
        loss.backward()
        res.append((float(loss.detach()), {k: p.grad.detach().cpu() for k, p in net.named_parameters() if p.grad is not None}))
    assert trainable.FALLBACKS == {}, trainable.FALLBACKS
    assert res[1][0] == pytest.approx(res[0][0], rel=5e-5)
    assert res[0][1].keys() == res[1][1].keys()
    for k in ("init_conv.weight", "init_conv.bias", "pos_enc.weights.weight", "downs.1.2.norm2.weight", "downs.0.0.block1.proj.weight", "final_conv.weight"):
        ref, got = res[0][1][k], res[1][1][k]
        assert rel_err(got.numpy(), ref.numpy()) <= 3e-4 * max(1.0, float(ref.abs().max())), k


@pytest.mark.gpu
def test_data_parallel_replicas_train_through_their_broadcast_parameters():
    """nn.DataParallel.forward over several device_ids (models/modules.py:81, the trainer's wrapper) replicates the module: a replica holds no
    Parameters, only the broadcast copies ``replicate`` leaves in ``_former_parameters``.  Under autograd the replicas evaluate the graph over those
    copies (net.py::_parameter_table), and the gradients that arrive at the owner's parameters equal the bare module's.  (Two logical replicas on
    cuda:0: the box has one card.)"""
    from noisediff_amd import NoiseDiffNet
    dev = torch.device("cuda", 0)
    x0, noise, t, cond = _inputs()
    x0, noise, t = x0.to(dev), noise.to(dev), t.to(dev)
    cond_dev = {k: v.to(dev) for k, v in cond.items()}
    grads = []
    for wrap in (False, True):
        net = NoiseDiffNet(SimpleNamespace(dim=DIM, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False, phase="train"))
        net.load_state_dict(state_dict(DIM), strict=True)
        net = net.to(dev).train()
        model = nn.DataParallel(net, device_ids=[0, 0]) if wrap else net
        gd = GaussianDiffusion(model, image_size=H, timesteps=T, beta_schedule="sigmoid2", objective="pred_v").to(dev)
        loss = gd.p_losses(x0, t, cond_dev, noise=noise.clone())
        loss.backward()
        grads.append((float(loss.detach()), {k: p.grad.detach().cpu() for k, p in net.named_parameters()}))
    assert grads[1][0] == pytest.approx(grads[0][0], rel=1e-5)
    assert grads[1][0] == pytest.approx(float(golden_value("train.loss.pred_v")), rel=5e-5)
    for k, g in grads[0][1].items():
        assert rel_err(grads[1][1][k].numpy(), g.numpy()) < 5e-4, k


@pytest.mark.gpu
def test_a_whole_training_step_captures_into_one_graph():
    """forward + backward + Adam of the .hip() network as one torch.cuda.CUDAGraph: the library launches on torch's capture stream.
    Replays keep training: the loss of a fixed batch goes down and the weights move."""
    dev = torch.device("cuda", 0)
    x0, noise, t, cond = _inputs()
    x0, noise, t = x0.to(dev), noise.to(dev), t.to(dev)
    cond = {k: v.to(dev) for k, v in cond.items()}

    def make():
        net = _net().to(dev).hip()
        gd = GaussianDiffusion(net, image_size=H, timesteps=T, beta_schedule="sigmoid2", objective="pred_v").to(dev)
        return net, gd, torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True)

    net, gd, opt = make()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                      # warm-up outside the capture: lazy library / optimizer state initialisation
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            gd.p_losses(x0, t, cond, noise=noise.clone()).backward()
            opt.step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    start = {k: v.detach().clone() for k, v in net.state_dict().items()}
    g = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        loss = gd.p_losses(x0, t, cond, noise=noise.clone())
        loss.backward()
        opt.step()
    losses = []
    for _ in range(4):
        g.replay()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert any(not torch.equal(v, start[k]) for k, v in net.state_dict().items())


def test_the_cached_parameter_table_follows_replaced_parameters_and_is_not_inherited_by_shallow_copies():
    """TrainableNoiseDiffNet.forward keeps name -> Parameter in the module's __dict__ (named_parameters() per step was 1.5 ms of host time).  It must notice a
    replaced Parameter, and an nn.DataParallel replica -- a shallow copy of the owner's __dict__ holding its own broadcast tensors -- must build its own."""
    import copy
    net = TrainableNoiseDiffNet(SimpleNamespace(dim=16))
    x0, noise, t, cond = _inputs()
    with torch.no_grad():
        y0 = net(x0, t, cond)
        table = net.__dict__["_nd_param_table"]
        assert net(x0, t, cond).equal(y0) and net.__dict__["_nd_param_table"] is table            # reused
        net.final_conv.weight = nn.Parameter(torch.zeros_like(net.final_conv.weight))
        net.final_conv.bias = nn.Parameter(torch.full_like(net.final_conv.bias, 0.25))
        y1 = net(x0, t, cond)
    assert net.__dict__["_nd_param_table"] is not table
    assert not y1.equal(y0)
    replica = copy.copy(net)                                                                    # what torch.nn.parallel.replicate starts from
    replica.__dict__ = net.__dict__.copy()
    with torch.no_grad():
        replica(x0, t, cond)
    assert replica.__dict__["_nd_param_table"][2] == id(replica) and net.__dict__["_nd_param_table"][2] == id(net)
