#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (build container only).

    python tests/golden/capture_golden.py            # writes fixtures next to this file

/root/reference is imported read-only with ``sys.dont_write_bytecode``; two imports it
makes but never uses on the sampling path (torchvision, ema_pytorch) are stubbed with
empty modules.  Weights / conditions / noise come from ``noisediff_amd.synth`` (hash
streams keyed by tensor name), so fixtures hold *outputs only* and stay small.
Nothing of the reference is written anywhere: the fixtures are arrays of numbers.

The GPU box has no /root/reference; tests read the fixtures, never this script's imports.
"""
from __future__ import annotations

import json
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

from noisediff_amd import synth  # noqa: E402
from noisediff_amd.spec import attention_param_spec, noisediff_param_spec  # noqa: E402


def import_reference():
    for name in ("torchvision", "torchvision.transforms", "torchvision.utils", "ema_pytorch"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    sys.modules["ema_pytorch"].EMA = object
    sys.path.insert(0, REF)
    import models.denoising_diffusion_pytorch as ddp
    import models.archs.Diffusion_arch as arch
    return ddp, arch


def ref_net(arch, dim, seed=0):
    args = SimpleNamespace(dim=dim, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False)
    net = arch.NoiseDiffNet(args).eval()
    sd = synth.make_state_dict(noisediff_param_spec(dim), seed)
    net.load_state_dict(sd, strict=True)          # proves the spec's names and shapes
    return net


class PatchedNoise:
    """Replace torch.randn / randn_like by the named hash streams for the duration of sample()."""

    def __init__(self, seed, batch, channels, size):
        self.seed, self.b, self.c, self.s = seed, batch, channels, size
        self.calls = []

    def __enter__(self):
        self._randn, self._randn_like = torch.randn, torch.randn_like
        self.first = True
        self.draw = 0

        def randn(shape, *a, **k):
            assert self.first, "torch.randn called twice"
            self.first = False
            self.calls.append("x_T")
            return synth.make_noise(self.seed, "x_T", self.b, self.c, self.s)

        def randn_like(t, *a, **k):
            name = f"noise.{self.draw}"
            self.draw += 1
            self.calls.append(name)
            return synth.make_noise(self.seed, name, self.b, self.c, self.s)

        torch.randn, torch.randn_like = randn, randn_like
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like = self._randn, self._randn_like


def sub(t: torch.Tensor, n: int = 4096) -> np.ndarray:
    """Deterministic strided subsample of a tensor for tap fixtures."""
    f = t.detach().reshape(-1)
    step = max(f.numel() // n, 1)
    return f[::step][:n].numpy().copy()


def capture_schedules(ddp, out):
    dummy = torch.nn.DataParallel(torch.nn.Identity())
    dummy.module.channels = dummy.module.out_dim = 4
    dummy.module.random_or_learned_sinusoidal_cond = False
    dummy.module.self_condition = False
    names = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
             "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
             "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
             "posterior_mean_coef1", "posterior_mean_coef2", "loss_weight"]
    for sched in ("linear", "cosine", "sigmoid1", "sigmoid2", "sigmoid3"):
        for T in (1000, 20):
            gd = ddp.GaussianDiffusion(dummy, image_size=8, timesteps=T, beta_schedule=sched, objective="pred_v")
            out[f"sched.{sched}.{T}"] = np.stack([getattr(gd, n).numpy() for n in names])
    out["sched.names"] = np.array(names)
    for T, S in ((1000, 50), (1000, 250), (20, 5), (1000, 999)):
        times = torch.linspace(-1, T - 1, steps=S + 1)
        out[f"ddim_times.{T}.{S}"] = np.array(list(reversed(times.int().tolist())), dtype=np.int64)
    gd = ddp.GaussianDiffusion(dummy, image_size=8, timesteps=1000, beta_schedule="sigmoid2", objective="pred_noise")
    out["sched.sigmoid2.1000.loss_weight.pred_noise"] = gd.loss_weight.numpy()
    gd = ddp.GaussianDiffusion(dummy, image_size=8, timesteps=1000, beta_schedule="sigmoid2", objective="pred_x0")
    out["sched.sigmoid2.1000.loss_weight.pred_x0"] = gd.loss_weight.numpy()


def capture_modules(arch, out):
    """Leaf / module goldens at d=16, B=2, 16x16 -- inputs are synth streams named 'mod.<x>'."""
    dim, B, H = 16, 2, 16
    net = ref_net(arch, dim)
    x = synth.uniform(7, "mod.x", (B, dim, H, H), -1.5, 1.5)
    x2 = synth.uniform(7, "mod.x2", (B, 2 * dim, H, H), -1.5, 1.5)
    x8 = synth.uniform(7, "mod.x8", (B, 8, H, H), -1.0, 1.0)
    temb = synth.uniform(7, "mod.temb", (B, 4 * dim), -1.0, 1.0)
    pos = synth.make_position(B, H, seed=7)
    iso_idx = synth.randint(7, "mod.iso", (B,), 0, 75)
    time = torch.tensor([0, 999], dtype=torch.long)
    with torch.no_grad():
        iso = net.iso_embed(iso_idx).unsqueeze(1)
        pos_emb = net.pos_mlp(net.pos_enc(pos))
        out["mod.pos_enc"] = net.pos_enc(pos).numpy()
        out["mod.pos_emb"] = pos_emb.numpy()
        out["mod.time_mlp"] = net.time_mlp(time).numpy()
        out["mod.time_mlp_mid"] = net.time_mlp(torch.tensor([1, 500])).numpy()
        out["mod.block"] = net.downs[0][0].block1(x).numpy()
        ss = (synth.uniform(7, "mod.scale", (B, dim, 1, 1), -0.5, 0.5), synth.uniform(7, "mod.shift", (B, dim, 1, 1), -0.5, 0.5))
        out["mod.block_ss"] = net.downs[0][0].block1(x, scale_shift=ss).numpy()
        out["mod.resnet_same"] = net.downs[0][0](x, temb).numpy()                     # 16 -> 16, G=8
        out["mod.resnet_resconv"] = net.final_res_block(x2, temb).numpy()             # 32 -> 16 (res_conv)
        out["mod.resnet_g2"] = net.shot_time(x, temb).numpy()                         # G=2 (ks=1 quirk => 3x3)
        out["mod.resnet_pos"] = net.pos_block1(x, pos_emb).numpy()
        out["mod.attn_block"] = net.downs[0][2](x, iso).numpy()
        out["mod.mlp_shot1"] = net.shot_mlp1(x8).numpy()
        out["mod.mlp_shot3"] = net.shot_mlp3(x).numpy()
        out["mod.downsample"] = net.downs[0][3](x).numpy()                            # unshuffle + 1x1 (16*4 -> 16)
        out["mod.upsample"] = net.ups[2][3](synth.uniform(7, "mod.xu", (B, 2 * dim, H // 2, H // 2), -1.5, 1.5)).numpy()  # 32 -> 16
        out["mod.init_conv"] = net.init_conv(synth.uniform(7, "mod.x4", (B, 4, H, H), -1.5, 1.5)).numpy()
        out["mod.final_conv"] = net.final_conv(x).numpy()
        # unwired attention classes (BASELINE config 4 / north-star extension)
        C = 8 * dim
        xa = synth.uniform(7, "mod.xa", (B, C, 8, 8), -1.5, 1.5)
        att = arch.Attention(C, heads=4, dim_head=32, flash=False).eval()
        sda = synth.make_state_dict(attention_param_spec("mid_attn", C), 0)
        att.load_state_dict({k[len("mid_attn."):]: v for k, v in sda.items()}, strict=True)
        out["mod.attention"] = att(xa).numpy()
        att_f = arch.Attention(C, heads=4, dim_head=32, flash=True).eval()
        att_f.load_state_dict(att.state_dict())
        out["mod.attention_flash"] = att_f(xa).numpy()
        lat = arch.LinearAttention(C, heads=4, dim_head=32).eval()
        sdl = {"norm.g": torch.ones(1, C, 1, 1),
               "to_qkv.weight": sda["mid_attn.to_qkv.weight"],
               "to_out.0.weight": sda["mid_attn.to_out.weight"], "to_out.0.bias": sda["mid_attn.to_out.bias"],
               "to_out.1.g": synth.uniform(7, "mod.lat_g", (1, C, 1, 1), 0.5, 1.5)}
        lat.load_state_dict(sdl, strict=True)
        out["mod.linear_attention"] = lat(xa).numpy()
        out["mod.rmsnorm"] = arch.RMSNorm(C)(xa).numpy()


def forward_with_taps(net, x, t, cond):
    """One reference forward; module outputs recorded by forward hooks."""
    taps = {}
    hooks = []

    def rec(name):
        def fn(_m, _i, o):
            taps[name] = o
        return fn

    named = {"pos_block1": net.pos_block1, "mid": net.mid_block2, "read_noise": net.final_conv,
             "shot_noise": net.shot_mlp3, "t_emb": net.time_mlp, "pos_emb": net.pos_mlp}
    for i in range(4):
        named[f"down{i}"] = net.downs[i][3]
        named[f"up{i}"] = net.ups[i][3]
    for k, m in named.items():
        hooks.append(m.register_forward_hook(rec(k)))
    with torch.no_grad():
        y = net(x, t, cond)
    for h in hooks:
        h.remove()
    return y, taps


def capture_net(arch, out):
    for dim, B, H in ((16, 2, 32), (32, 2, 64)):
        net = ref_net(arch, dim)
        cond = synth.make_condition(B, H, seed=1)
        x = synth.make_noise(2, "net.x", B, 4, H)
        for t in (0, 500, 999):
            y, taps = forward_with_taps(net, x, torch.full((B,), t, dtype=torch.long), cond)
            out[f"net.d{dim}.h{H}.t{t}"] = y.numpy()
            if t == 500:
                for k, v in taps.items():
                    out[f"net.d{dim}.h{H}.t{t}.tap.{k}"] = sub(v)
        # per-sample timesteps (the training-style call of forward)
        tt = torch.tensor([3, 777], dtype=torch.long)
        y, _ = forward_with_taps(net, x, tt, cond)
        out[f"net.d{dim}.h{H}.tmixed"] = y.numpy()


def run_sampler(ddp, arch, dim, B, H, T, S, eta, return_all, v_steps, mid_attn=False, sched="sigmoid2",
                objective="pred_v", preset=False):
    net = ref_net(arch, dim)
    hooks = []
    if mid_attn:
        C = 8 * dim
        att = arch.Attention(C, heads=4, dim_head=32, flash=False).eval()
        sda = synth.make_state_dict(attention_param_spec("mid_attn", C), 0)
        att.load_state_dict({k[len("mid_attn."):]: v for k, v in sda.items()}, strict=True)
        hooks.append(net.mid_block1.register_forward_hook(lambda _m, _i, o: att(o) + o))
    wrapped = torch.nn.DataParallel(net)          # bare modules crash at denoising_diffusion_pytorch.py:189
    gd = ddp.GaussianDiffusion(wrapped, image_size=H, timesteps=T, sampling_timesteps=S, beta_schedule=sched,
                               objective=objective, ddim_sampling_eta=eta)
    cond = synth.make_condition(B, H, seed=1)
    vs = {}
    calls = {"n": 0}

    def grab(_m, _i, o):
        if calls["n"] in v_steps:
            vs[calls["n"]] = o.numpy().copy()
        calls["n"] += 1

    hooks.append(net.register_forward_hook(grab))
    with PatchedNoise(2, B, 4, H) as pn:
        kw = {}
        if preset:
            kw["preset_mean"] = synth.make_noise(2, "x_T", B, 4, H)
        res = gd.sample(batch_size=B, condition=cond, return_all_timesteps=return_all, **kw)
    for h in hooks:
        h.remove()
    return res.numpy(), vs, pn.calls


def capture_sampler(ddp, arch, out, meta):
    # BASELINE config 1 exactly: d=32, 64x64x4, 50-step DDIM (eta 0), batch 4
    res, vs, calls = run_sampler(ddp, arch, 32, 4, 64, 1000, 50, 0.0, False, {0, 25, 49})
    out["samp.cfg1.out"] = res
    for k, v in vs.items():
        out[f"samp.cfg1.v{k}"] = v
    meta["samp.cfg1.calls"] = calls
    # 20-step DDPM exercising the noise path (+ preset_mean variant: x_T through the reference's own hook)
    res, vs, calls = run_sampler(ddp, arch, 16, 2, 32, 20, None, 0.0, False, {0, 10, 19})
    out["samp.ddpm20.out"] = res
    for k, v in vs.items():
        out[f"samp.ddpm20.v{k}"] = v
    meta["samp.ddpm20.calls"] = calls
    res, _, _ = run_sampler(ddp, arch, 16, 2, 32, 20, None, 0.0, False, set(), preset=True)
    out["samp.ddpm20_preset.out"] = res
    # return_all_timesteps on a T=4 toy
    res, _, calls = run_sampler(ddp, arch, 16, 2, 32, 4, None, 0.0, True, set())
    out["samp.ddpm4_all.out"] = res
    meta["samp.ddpm4_all.calls"] = calls
    # DDIM with eta > 0 (sigma * noise term) and return_all
    res, _, calls = run_sampler(ddp, arch, 16, 2, 32, 20, 5, 0.5, True, set())
    out["samp.ddim5_eta.out"] = res
    meta["samp.ddim5_eta.calls"] = calls
    # other objectives / schedule through the same code
    # (linear at T=20 would hit beta=1 => 1/alphas_cumprod = inf in the reference itself; use T=50)
    res, _, _ = run_sampler(ddp, arch, 16, 2, 32, 50, None, 0.0, False, set(), sched="linear", objective="pred_noise")
    out["samp.ddpm50_eps_linear.out"] = res
    res, _, _ = run_sampler(ddp, arch, 16, 2, 32, 20, 5, 0.0, False, set(), sched="cosine", objective="pred_x0")
    out["samp.ddim5_x0_cosine.out"] = res
    # config-4 extension: Attention spliced between mid_block1 and mid_block2 (d=16, 64x64 -> 64 tokens, C=128)
    res, vs, _ = run_sampler(ddp, arch, 16, 2, 64, 1000, 10, 0.0, False, {0, 9}, mid_attn=True)
    out["samp.cfg4toy.out"] = res
    for k, v in vs.items():
        out[f"samp.cfg4toy.v{k}"] = v


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ddp, arch = import_reference()
    meta = {"torch": torch.__version__}

    for dim in (16, 32, 48, 64):
        args = SimpleNamespace(dim=dim, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False)
        sd = arch.NoiseDiffNet(args).state_dict()
        meta[f"state_dict.d{dim}"] = [[k, list(v.shape)] for k, v in sd.items()]

    groups = {"schedules": capture_schedules, "modules": capture_modules, "net": capture_net}
    for name, fn in groups.items():
        out = {}
        fn(ddp, out) if name == "schedules" else fn(arch, out)
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        print(name, len(out), "arrays")
    out = {}
    capture_sampler(ddp, arch, out, meta)
    np.savez_compressed(os.path.join(HERE, "sampler.npz"), **out)
    print("sampler", len(out), "arrays")
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f)


if __name__ == "__main__":
    main()
