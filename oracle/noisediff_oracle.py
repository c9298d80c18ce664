"""CPU oracle for the NoiseDiff sampling hot path.  TEST INFRASTRUCTURE ONLY.

This is a functional restatement (plain torch CPU ops on a flat state-dict, fp32,
NCHW like the reference) of ``GaussianDiffusion.sample`` -> ``p_sample_loop`` /
``ddim_sample`` -> ``NoiseDiffNet.forward``.  It exists so that the HIP path can be
checked on a box where /root/reference does not exist.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it;
nothing under ``noisediff_amd/`` does, and the product path raises if its HIP
library is missing rather than falling back to anything here.

Parity status: PINNED.  ``tests/golden/capture_golden.py`` imports the real
reference in the build container and stores its outputs for schedules, every leaf
module, whole-network forwards and end-to-end DDIM/DDPM runs; ``tests/test_oracle.py``
checks this file against those fixtures (max-abs <= 2e-5 on O(1) activations).

Every function cites the reference lines it restates (paths relative to
/root/reference).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

# --------------------------------------------------------------------------- schedules


def _sigmoid64(x: np.ndarray) -> np.ndarray:
    return 1.0 / (1.0 + np.exp(-x))


def beta_schedule(name: str, timesteps: int) -> np.ndarray:
    """float64 betas.  models/denoising_diffusion_pytorch.py:96-164, selection :206-218."""
    if name == "linear":                                   # :96-103
        scale = 1000.0 / timesteps
        return np.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=np.float64)
    t = np.linspace(0, timesteps, timesteps + 1, dtype=np.float64) / timesteps
    if name == "cosine":                                   # :105-115
        s = 0.008
        ac = np.cos((t + s) / (1 + s) * math.pi * 0.5) ** 2
    elif name in ("sigmoid1", "sigmoid2", "sigmoid3"):     # :119-164
        start, end, tau = {"sigmoid1": (-3, 3, 0.5), "sigmoid2": (-7, 3, 0.7),
                           "sigmoid3": (-10, 3, 0.7)}[name]
        # the reference builds v_start/v_end from float32 scalars: torch.tensor(start / tau)
        v_start = float(torch.tensor(start / tau).sigmoid())
        v_end = float(torch.tensor(end / tau).sigmoid())
        ac = (-_sigmoid64((t * (end - start) + start) / tau) + v_end) / (v_end - v_start)
    else:
        raise ValueError(f"unknown beta schedule {name}")   # :218
    ac = ac / ac[0]
    return np.clip(1 - ac[1:] / ac[:-1], 0, 0.999)


def schedule_buffers(name: str, timesteps: int, objective: str = "pred_v") -> Dict[str, np.ndarray]:
    """The 13 fp32 buffers of GaussianDiffusion.__init__ (:220-286)."""
    np.seterr(divide="ignore", invalid="ignore")   # linear at tiny T reaches beta = 1 exactly as the reference does
    betas = beta_schedule(name, timesteps)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas)
    ac_prev = np.concatenate([[1.0], ac[:-1]])              # F.pad(..., value=1.)  :225
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)         # :256
    snr = ac / (1 - ac)                                     # :275
    lw = {"pred_noise": snr / snr, "pred_x0": snr, "pred_v": snr / (snr + 1)}[objective]
    f64 = {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": np.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": np.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": np.log(1.0 - ac),
        "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / ac - 1),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": np.log(np.maximum(post_var, 1e-20)),   # :264
        "posterior_mean_coef1": betas * np.sqrt(ac_prev) / (1.0 - ac),          # :265
        "posterior_mean_coef2": (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac), # :266
        "loss_weight": lw,
    }
    return {k: v.astype(np.float32) for k, v in f64.items()}


def ddim_time_pairs(total: int, sampling: int) -> List[Tuple[int, int]]:
    """:409-411 -- linspace(-1, T-1, S+1) computed in fp32, truncated, reversed, paired."""
    times = torch.linspace(-1, total - 1, steps=sampling + 1)
    times = list(reversed(times.int().tolist()))
    return list(zip(times[:-1], times[1:]))


# --------------------------------------------------------------------------- leaf modules


def conv(sd: SD, p: str, x: torch.Tensor, padding: int = 0) -> torch.Tensor:
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), padding=padding)


def linear(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def block(sd: SD, p: str, x: torch.Tensor, groups: int,
          scale_shift: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> torch.Tensor:
    """Block.forward  models/archs/Diffusion_arch.py:135-144 (conv3x3 -> GN -> mod -> SiLU)."""
    x = conv(sd, p + ".proj", x, padding=1)
    x = F.group_norm(x, groups, sd[p + ".norm.weight"], sd[p + ".norm.bias"], eps=1e-5)
    if scale_shift is not None:
        scale, shift = scale_shift
        x = x * (scale + 1) + shift
    return F.silu(x)


def _res(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return conv(sd, p + ".res_conv", x) if (p + ".res_conv.weight") in sd else x


def resnet_block(sd: SD, p: str, x: torch.Tensor, temb: Optional[torch.Tensor], groups: int) -> torch.Tensor:
    """ResnetBlock.forward :158-170 -- time MLP gives per-(sample, channel) scale/shift."""
    ss = None
    if temb is not None:
        e = linear(sd, p + ".mlp.1", F.silu(temb))[:, :, None, None]
        ss = e.chunk(2, dim=1)
    h = block(sd, p + ".block1", x, groups, ss)
    h = block(sd, p + ".block2", h, groups)
    return h + _res(sd, p, x)


def resnet_block_pos(sd: SD, p: str, x: torch.Tensor, pos_emb: torch.Tensor, groups: int) -> torch.Tensor:
    """ResnetBlock2.forward :185-196 -- scale/shift are per-pixel maps from pos_emb."""
    e = conv(sd, p + ".mlp.1", F.silu(pos_emb))
    ss = e.chunk(2, dim=1)
    h = block(sd, p + ".block1", x, groups, ss)
    h = block(sd, p + ".block2", h, groups)
    return h + _res(sd, p, x)


def mlp(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """Mlp.forward :350-356 with act_layer=nn.GELU (exact erf), dropout p=0."""
    return conv(sd, p + ".fc2", F.gelu(conv(sd, p + ".fc1", x)))


def cross_attention(sd: SD, p: str, x: torch.Tensor, ctx: torch.Tensor, heads: int = 4) -> torch.Tensor:
    """CrossAttention.forward :379-402, written out in full (no 1-token shortcut here)."""
    b, n, _ = x.shape
    q = linear(sd, p + ".to_q", x)
    k = linear(sd, p + ".to_k", ctx)
    v = linear(sd, p + ".to_v", ctx)
    d = q.shape[-1] // heads

    def split(t: torch.Tensor) -> torch.Tensor:       # 'b n (h d) -> (b h) n d'  :387
        return t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3).reshape(b * heads, t.shape[1], d)

    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum("bid,bjd->bij", q, k) * (d ** -0.5)         # :389
    attn = sim.softmax(dim=-1)                                     # :398
    out = torch.einsum("bij,bjd->bid", attn, v)                    # :400
    out = out.reshape(b, heads, n, d).permute(0, 2, 1, 3).reshape(b, n, heads * d)   # :401
    return linear(sd, p + ".to_out.0", out)


def attn_block(sd: SD, p: str, x: torch.Tensor, ctx: torch.Tensor) -> torch.Tensor:
    """AttnBlock.forward :434-443."""
    b, c, h, w = x.shape
    x_in = x
    t = x.permute(0, 2, 3, 1).reshape(b, h * w, c)                 # 'b c h w -> b (h w) c'
    n1 = F.layer_norm(t, (c,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], eps=1e-5)
    t = cross_attention(sd, p + ".attn", n1, ctx) + t              # :438
    n2 = F.layer_norm(t, (c,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], eps=1e-5)
    ff = linear(sd, p + ".ff.net.2", F.gelu(linear(sd, p + ".ff.net.0.0", n2)))   # :410-419
    t = ff + t                                                     # :439
    t = t.reshape(b, h, w, c).permute(0, 3, 1, 2)
    return conv(sd, p + ".proj_out", t) + x_in                     # :441-443


def learned_sinusoidal_pos_emb(sd: SD, p: str, position: torch.Tensor) -> torch.Tensor:
    """LearnedSinusoidalPosEmb.forward :331-337 -- cat(w, sin 2*pi*w, cos 2*pi*w)."""
    w = conv(sd, p + ".weights", position)
    f = w * 2 * math.pi
    return torch.cat((w, f.sin(), f.cos()), dim=1)


def sinusoidal_pos_emb(time: torch.Tensor, dim: int, theta: float = 10000.0) -> torch.Tensor:
    """SinusoidalPosEmb.forward :100-107."""
    half = dim // 2
    e = math.log(theta) / (half - 1)
    e = torch.exp(torch.arange(half) * -e)
    e = time[:, None] * e[None, :]            # int64 * fp32 -> fp32
    return torch.cat((e.sin(), e.cos()), dim=-1)


def time_mlp(sd: SD, time: torch.Tensor, dim: int) -> torch.Tensor:
    """time_mlp :502-507 -- sinusoidal -> Linear -> GELU -> Linear."""
    e = sinusoidal_pos_emb(time, dim)
    return linear(sd, "time_mlp.3", F.gelu(linear(sd, "time_mlp.1", e)))


def pixel_unshuffle_conv(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """Downsample :78-82 -- 'b c (h p1) (w p2) -> b (c p1 p2) h w' then conv1x1."""
    b, c, hh, ww = x.shape
    t = x.reshape(b, c, hh // 2, 2, ww // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(b, c * 4, hh // 2, ww // 2)
    return conv(sd, p + ".1", t)


def upsample_conv(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """Upsample :72-76 -- nearest x2 then conv3x3."""
    t = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    return conv(sd, p + ".1", t, padding=1)


def rms_norm(g: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """RMSNorm.forward :89-90 -- F.normalize over channels (eps 1e-12) * g * sqrt(C)."""
    n = x.norm(dim=1, keepdim=True).clamp_min(1e-12)
    return x / n * g * (x.shape[1] ** 0.5)


def attention(sd: SD, p: str, x: torch.Tensor, heads: int = 4) -> torch.Tensor:
    """Attention.forward :255-266 with Attend's explicit path (models/attend.py:101-116)."""
    b, c, h, w = x.shape
    xn = rms_norm(sd[p + ".norm.g"], x)
    qkv = F.conv2d(xn, sd[p + ".to_qkv.weight"]).chunk(3, dim=1)
    q, k, v = (t.reshape(b, heads, -1, h * w).permute(0, 1, 3, 2) for t in qkv)   # b h (xy) c
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * (q.shape[-1] ** -0.5)
    out = torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), v)
    out = out.permute(0, 1, 3, 2).reshape(b, -1, h, w)                            # b (h d) x y
    return conv(sd, p + ".to_out", out)


def linear_attention(sd: SD, p: str, x: torch.Tensor, heads: int = 4) -> torch.Tensor:
    """LinearAttention.forward :218-235."""
    b, c, h, w = x.shape
    xn = rms_norm(sd[p + ".norm.g"], x)
    qkv = F.conv2d(xn, sd[p + ".to_qkv.weight"]).chunk(3, dim=1)
    q, k, v = (t.reshape(b, heads, -1, h * w) for t in qkv)                       # b h c (xy)
    q = q.softmax(dim=-2) * (q.shape[2] ** -0.5)
    k = k.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(b, -1, h, w)
    out = conv(sd, p + ".to_out.0", out)
    return rms_norm(sd[p + ".to_out.1.g"], out)


# --------------------------------------------------------------------------- the network


def net_dim(sd: SD) -> int:
    return sd["init_conv.weight"].shape[0]


def noisediff_forward(sd: SD, x: torch.Tensor, time: torch.Tensor, condition: Dict[str, torch.Tensor],
                      mid_attention: Optional[str] = None,
                      taps: Optional[Dict[str, torch.Tensor]] = None, stage_attention=None) -> torch.Tensor:
    """NoiseDiffNet.forward  models/archs/Diffusion_arch.py:577-646.

    ``mid_attention``: state-dict prefix of an ``Attention`` block applied as
    ``x = attn(x) + x`` between mid_block1 and mid_block2 (BASELINE config 4; the
    reference computes ``FullAttention`` at :518 but never wires it).
    ``taps``: optional dict that receives named intermediates for module-level tests.
    ``stage_attention``: four entries 'linear' / 'full' / None -- upstream's per-stage wiring of the ``LinearAttention`` / ``Attention``
    classes the reference defines and drops (:198-266; flags computed at :467-468,509-518): ``x = attn(x) + x`` behind a stage's second
    ResnetBlock (in front of the skip), modules ``down_attns.{i}`` / ``up_attns.{i}`` (up stage i uses entry 3 - i).  Pinned by
    tests/golden/stage_attn.npz (tests/golden/capture_stage_attn.py: forward hooks on the reference net's blocks).
    """
    dim = net_dim(sd)

    def stage_attn(prefix: str, kind, v: torch.Tensor) -> torch.Tensor:
        if not kind:
            return v
        return (attention(sd, prefix, v) if kind == "full" else linear_attention(sd, prefix, v)) + v

    kinds = tuple(stage_attention) if stage_attention else (None,) * 4
    assert x.shape[-1] % 8 == 0 and x.shape[-2] % 8 == 0                      # :578
    clean, position = condition["clean_img"], condition["position"]
    G = 8

    def tap(name: str, t: torch.Tensor) -> torch.Tensor:
        if taps is not None:
            taps[name] = t
        return t

    pos_emb = mlp(sd, "pos_mlp", learned_sinusoidal_pos_emb(sd, "pos_enc", position))   # :584-585
    tap("pos_emb", pos_emb)
    iso = F.embedding(condition["iso_ratio_idx"].long(), sd["iso_embed.weight"]).unsqueeze(1)   # :591
    t = time_mlp(sd, time, dim)                                                # :595
    tap("t_emb", t)

    s = mlp(sd, "shot_mlp1", torch.cat([clean, x], dim=1))                     # :598
    r_shot = s
    s = attn_block(sd, "shot_attn", s, iso)                                    # :600
    s = mlp(sd, "shot_mlp2", s)
    s = resnet_block(sd, "shot_time", s, t, groups=2) + r_shot                 # :602-603
    shot_noise = tap("shot_noise", mlp(sd, "shot_mlp3", s))                    # :604

    x = conv(sd, "init_conv", x, padding=3)                                    # :606
    r = x
    hs: List[torch.Tensor] = []
    x = tap("pos_block1", resnet_block_pos(sd, "pos_block1", x, pos_emb, groups=2))   # :611

    for i in range(4):                                                         # :613-622
        p = f"downs.{i}"
        x = resnet_block(sd, p + ".0", x, t, G); hs.append(x)
        x = stage_attn(f"down_attns.{i}", kinds[i], resnet_block(sd, p + ".1", x, t, G)); hs.append(x)
        x = attn_block(sd, p + ".2", x, iso)
        x = conv(sd, p + ".3", x, padding=1) if i == 3 else pixel_unshuffle_conv(sd, p + ".3", x)
        tap(f"down{i}", x)

    x = resnet_block(sd, "mid_block1", x, t, G)                                # :624
    if mid_attention is not None:
        x = attention(sd, mid_attention, x) + x
    x = tap("mid", resnet_block(sd, "mid_block2", x, t, G))                    # :625

    for i in range(4):                                                         # :627-636
        p = f"ups.{i}"
        x = resnet_block(sd, p + ".0", torch.cat((x, hs.pop()), dim=1), t, G)
        x = stage_attn(f"up_attns.{i}", kinds[3 - i], resnet_block(sd, p + ".1", torch.cat((x, hs.pop()), dim=1), t, G))
        x = attn_block(sd, p + ".2", x, iso)
        x = conv(sd, p + ".3", x, padding=1) if i == 3 else upsample_conv(sd, p + ".3", x)
        tap(f"up{i}", x)

    x = resnet_block_pos(sd, "pos_block2", x, pos_emb, groups=2)               # :638
    x = resnet_block(sd, "final_res_block", torch.cat((x, r), dim=1), t, G)    # :640-642
    read_noise = tap("read_noise", conv(sd, "final_conv", x))                  # :643
    return shot_noise + read_noise                                             # :644


def posemb_unet_forward(sd: SD, arch: str, x: torch.Tensor, time: torch.Tensor, condition,
                        taps: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """The ``UNet_PosEmbV2*`` ablation nets, forward as in models/archs/others_arch.py:
    ``UNet_PosEmbV2`` :483-537, ``UNet_PosEmbV2_NoPosition`` :655-707 (``condition`` is the clean image itself, :658;
    a dict with ``clean_img`` is accepted too), ``UNet_PosEmbV2_CameraCond`` :919-985.  Parity status: PINNED by
    ``tests/golden/variants.npz`` (captured from the reference by ``tests/golden/capture_variants.py``)."""
    dim = net_dim(sd)
    assert x.shape[-1] % 8 == 0 and x.shape[-2] % 8 == 0
    position_aware = arch != "UNet_PosEmbV2_NoPosition"
    camera = arch == "UNet_PosEmbV2_CameraCond"
    clean = condition["clean_img"] if isinstance(condition, dict) else condition
    G = 8

    def tap(name: str, t: torch.Tensor) -> torch.Tensor:
        if taps is not None:
            taps[name] = t
        return t

    pos_emb = mlp(sd, "pos_mlp", learned_sinusoidal_pos_emb(sd, "pos_enc", condition["position"])) if position_aware else None
    clean_emb = conv(sd, "cond_init_conv", clean, padding=3)                    # :491 / :662 / :928
    clean_emb = tap("clean_emb", resnet_block(sd, "cond_res_block1", clean_emb, None, G))
    iso = F.embedding(condition["iso_ratio_idx"].long(), sd["iso_embed.weight"]).unsqueeze(1) if camera else None   # :931-933
    x = conv(sd, "init_conv", x, padding=3)                                     # :495
    r = x
    x = tap("cond_concat", conv(sd, "cond_concat_conv", torch.cat([x, clean_emb], dim=1), padding=1))   # :498
    t = time_mlp(sd, time, dim)
    pos_block = (lambda p, v: resnet_block_pos(sd, p, v, pos_emb, groups=2)) if position_aware else \
                (lambda p, v: resnet_block(sd, p, v, None, groups=2))           # :644-646, :675
    x = tap("pos_block1", pos_block("pos_block1", x))
    rs = 3 if camera else 2
    hs: List[torch.Tensor] = []
    for i in range(4):                                                          # :507-515 / :947-957
        p = f"downs.{i}"
        x = resnet_block(sd, p + ".0", x, t, G); hs.append(x)
        x = resnet_block(sd, p + ".1", x, t, G); hs.append(x)
        if camera:
            x = attn_block(sd, p + ".2", x, iso)
        x = conv(sd, f"{p}.{rs}", x, padding=1) if i == 3 else pixel_unshuffle_conv(sd, f"{p}.{rs}", x)
        tap(f"down{i}", x)
    x = resnet_block(sd, "mid_block1", x, t, G)
    x = tap("mid", resnet_block(sd, "mid_block2", x, t, G))
    for i in range(4):                                                          # :520-527 / :962-971
        p = f"ups.{i}"
        x = resnet_block(sd, p + ".0", torch.cat((x, hs.pop()), dim=1), t, G)
        x = resnet_block(sd, p + ".1", torch.cat((x, hs.pop()), dim=1), t, G)
        if camera:
            x = attn_block(sd, p + ".2", x, iso)
        x = conv(sd, f"{p}.{rs}", x, padding=1) if i == 3 else upsample_conv(sd, f"{p}.{rs}", x)
        tap(f"up{i}", x)
    x = tap("pos_block2", pos_block("pos_block2", x))
    x = resnet_block(sd, "final_res_block", torch.cat((x, r), dim=1), t, G)     # :531-536
    return conv(sd, "final_conv", x)


# --------------------------------------------------------------------------- the sampler

NetFn = Callable[[torch.Tensor, torch.Tensor], torch.Tensor]
NoiseFn = Callable[[int, Sequence[int]], torch.Tensor]


def _coef(buf: Dict[str, np.ndarray], name: str, t: int) -> torch.Tensor:
    return torch.tensor(buf[name][t], dtype=torch.float32)      # extract() :91-94, uniform t


def predict_x0_eps(buf, objective: str, x: torch.Tensor, t: int, out: torch.Tensor, clip: bool):
    """model_predictions :331-354 after the network call."""
    def clipf(v):
        return v.clamp(-1.0, 1.0) if clip else v
    rc, rm1 = _coef(buf, "sqrt_recip_alphas_cumprod", t), _coef(buf, "sqrt_recipm1_alphas_cumprod", t)
    if objective == "pred_noise":
        eps = out
        x0 = clipf(rc * x - rm1 * eps)                          # :298-302
        if clip:
            eps = (rc * x - x0) / rm1                           # :340-341
    elif objective == "pred_x0":
        x0 = clipf(out)
        eps = (rc * x - x0) / rm1                               # :304-308
    elif objective == "pred_v":
        x0 = _coef(buf, "sqrt_alphas_cumprod", t) * x - _coef(buf, "sqrt_one_minus_alphas_cumprod", t) * out  # :316-320
        x0 = clipf(x0)
        eps = (rc * x - x0) / rm1
    else:
        raise ValueError(objective)
    return eps, x0


def p_sample_loop(net: NetFn, buf, objective: str, x_T: torch.Tensor, noise: NoiseFn,
                  return_all: bool = False, on_step=None) -> torch.Tensor:
    """p_sample_loop :375-402 with p_sample :366-373, p_mean_variance :356-364, q_posterior :322-329.

    ``noise(i, shape)`` supplies the i-th ``randn_like`` draw (one per step with t > 0).
    """
    T = len(buf["betas"])
    img = x_T
    imgs = [img]
    draw = 0
    for t in reversed(range(T)):
        tt = torch.full((img.shape[0],), t, dtype=torch.long)
        out = net(img, tt)
        if on_step is not None:
            on_step(t, img, out)
        _, x0 = predict_x0_eps(buf, objective, img, t, out, clip=False)
        x0 = x0.clamp(-1.0, 1.0)                                                      # :361
        mean = _coef(buf, "posterior_mean_coef1", t) * x0 + _coef(buf, "posterior_mean_coef2", t) * img
        if t > 0:
            z = noise(draw, img.shape); draw += 1
            img = mean + (0.5 * _coef(buf, "posterior_log_variance_clipped", t)).exp() * z   # :372
        else:
            img = mean + (0.5 * _coef(buf, "posterior_log_variance_clipped", t)).exp() * 0.0
        imgs.append(img)
    return torch.stack(imgs, dim=1) if return_all else img


def ddim_sample(net: NetFn, buf, objective: str, x_T: torch.Tensor, noise: NoiseFn,
                sampling_timesteps: int, eta: float = 0.0, return_all: bool = False,
                on_step=None) -> torch.Tensor:
    """ddim_sample :404-444.  ``noise(i, shape)`` = i-th randn_like (drawn even when eta == 0)."""
    T = len(buf["betas"])
    ac = torch.from_numpy(buf["alphas_cumprod"])
    img = x_T
    imgs = [img]
    draw = 0
    for time, time_next in ddim_time_pairs(T, sampling_timesteps):
        tt = torch.full((img.shape[0],), time, dtype=torch.long)
        out = net(img, tt)
        if on_step is not None:
            on_step(time, img, out)
        eps, x0 = predict_x0_eps(buf, objective, img, time, out, clip=True)           # :420
        if time_next < 0:
            img = x0
            imgs.append(img)
            continue
        a, an = ac[time], ac[time_next]
        sigma = eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()                      # :430
        c = (1 - an - sigma ** 2).sqrt()
        z = noise(draw, img.shape); draw += 1
        img = x0 * an.sqrt() + c * eps + sigma * z                                    # :435-437
        imgs.append(img)
    return torch.stack(imgs, dim=1) if return_all else img


def sample(sd: SD, condition: Dict[str, torch.Tensor], *, image_size: int, batch_size: int,
           timesteps: int = 1000, sampling_timesteps: Optional[int] = None,
           beta_schedule_name: str = "sigmoid2", objective: str = "pred_v", eta: float = 0.0,
           x_T: torch.Tensor, noise: NoiseFn, return_all: bool = False,
           mid_attention: Optional[str] = None, on_step=None, stage_attention=None) -> torch.Tensor:
    """GaussianDiffusion.sample :446-451 (auto_normalize=False => unnormalize is identity :290-291)."""
    buf = schedule_buffers(beta_schedule_name, timesteps, objective)
    S = timesteps if sampling_timesteps is None else sampling_timesteps
    assert S <= timesteps                                                             # :234
    assert tuple(x_T.shape) == (batch_size, sd["init_conv.weight"].shape[1], image_size, image_size)

    def net(x, t):
        return noisediff_forward(sd, x, t, condition, mid_attention=mid_attention, stage_attention=stage_attention)

    with torch.no_grad():
        if S < timesteps:                                                             # :235,449
            return ddim_sample(net, buf, objective, x_T, noise, S, eta, return_all, on_step)
        return p_sample_loop(net, buf, objective, x_T, noise, return_all, on_step)


# --------------------------------------------------------------------------- LSID denoiser (next row)


def lsid_forward(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """LSID.forward  models/archs/SID_arch.py:105-175: 4 x (conv3x3, LeakyReLU 0.2) x 2 + MaxPool(2, ceil), bottleneck,
    4 x (ConvTranspose 2x2 s2, crop, cat with the encoder feature, 2 convs), 1x1 head."""
    def cc(name, t):
        return F.leaky_relu(conv(sd, name, t, padding=1), 0.2)

    feats = []
    for i in range(1, 5):
        x = cc(f"conv{i}_2", cc(f"conv{i}_1", x))
        feats.append(x)
        x = F.max_pool2d(x, 2, 2, 0, ceil_mode=True)                                  # :60
    x = cc("conv5_2", cc("conv5_1", x))
    for i in range(6, 10):
        skip = feats.pop()
        x = F.conv_transpose2d(x, sd[f"up{i}.weight"], stride=2)
        x = torch.cat((x[:, :, :skip.shape[2], :skip.shape[3]], skip), 1)             # :135
        x = cc(f"conv{i}_2", cc(f"conv{i}_1", x))
    return conv(sd, "conv10", x)


def compose_and_denoise(sd_lsid: SD, noise: torch.Tensor, clean: torch.Tensor):
    """Synth -> denoise composition of BASELINE config 5: noisy = clip(clip(noise,-1,1) + clean, 0, 1)
    (dataloader/dataset_denoising.py:140-151), LSID forward, clamp to [0,1], PSNR = 10 log10(1/MSE)
    (test_denoising.py:220-226,334-343)."""
    noisy = (noise.clamp(-1.0, 1.0) + clean).clamp(0.0, 1.0)
    out = lsid_forward(sd_lsid, noisy).clamp(0.0, 1.0)
    mse = torch.mean((out.double() - clean.double()) ** 2).item()
    return noisy, out, 10.0 * math.log10(1.0 / mse)


# --------------------------------------------------------------------------- device RNG restatement


def philox4x32_10(counter: np.ndarray, key: np.ndarray) -> np.ndarray:
    """Philox-4x32-10 (Salmon et al., SC'11) on uint32 arrays: counter (..., 4), key (..., 2).

    Restates the device noise generator of ``nd_sampler_*`` (csrc/sampler.hip) so the
    throughput-mode noise stream can be checked bit-for-bit on the uniform integers.
    """
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
    c = [counter[..., i].astype(np.uint32) for i in range(4)]
    k0 = key[..., 0].astype(np.uint32)
    k1 = key[..., 1].astype(np.uint32)
    for _ in range(10):
        p0 = c[0].astype(np.uint64) * M0
        p1 = c[2].astype(np.uint64) * M1
        hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
        hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        with np.errstate(over="ignore"):
            k0 = (k0 + W0).astype(np.uint32)
            k1 = (k1 + W1).astype(np.uint32)
    return np.stack(c, axis=-1)


def philox_normal4(bits: np.ndarray) -> np.ndarray:
    """Four N(0,1) per Philox block: two Box-Muller pairs, u = (bits + 0.5) * 2^-32."""
    u = (bits.astype(np.float64) + 0.5) * (1.0 / 4294967296.0)
    r0 = np.sqrt(-2.0 * np.log(u[..., 0]))
    r1 = np.sqrt(-2.0 * np.log(u[..., 2]))
    a0 = 2.0 * math.pi * u[..., 1]
    a1 = 2.0 * math.pi * u[..., 3]
    return np.stack([r0 * np.cos(a0), r0 * np.sin(a0), r1 * np.cos(a1), r1 * np.sin(a1)], axis=-1)
