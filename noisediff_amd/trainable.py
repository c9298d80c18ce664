"""A differentiable NoiseDiffNet for training on MI355X without the reference tree.

    net = noisediff_amd.TrainableNoiseDiffNet(args).cuda().hip()          # same constructor argument as the registry's cls(args)
    loss = noisediff_amd.GaussianDiffusion(net, image_size=256, ...)(img, condition); loss.backward(); opt.step()
    noisediff_amd.NoiseDiffNet(args).load_state_dict(net.state_dict())   # the HIP sampler takes the trained weights as they are

The graph is the reference network's (models/archs/Diffusion_arch.py:447-646) written as one functional forward
over a flat parameter table; parameters are registered under the reference's own dotted names, so ``state_dict()`` is
interchangeable with the reference class and with ``noisediff_amd.NoiseDiffNet``.  ``.hip()`` routes every 3x3 convolution and
GroupNorm / LayerNorm through the HIP library forward and backward and the weight / bias gradients of the Linears and 1x1 convolutions
(noisediff_amd/train.py); everything else is PyTorch.  Parity: the
forward against the reference's golden activations and loss / gradients against tests/golden/training.npz (tests/test_trainable.py).
"""
from __future__ import annotations

import logging
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import synth
from .spec import noisediff_param_spec

P = Dict[str, torch.Tensor]
GROUPS, POS_GROUPS, HEADS = 8, 2, 4
_log = logging.getLogger(__name__)
FALLBACKS: Dict[Tuple[str, str], str] = {}      # (layer, operator) -> why it runs on PyTorch although the network is .hip() (filled on first use, logged once each)


class _ZeroGrads(torch.autograd.Function):
    """``y`` unchanged; the parameters after it receive exactly-zero gradients -- what autograd gives parameters whose only path to the loss runs through a
    softmax over ONE key (CrossAttention's to_q / to_k and the LayerNorm that feeds only them, one-token ISO context: SURVEY fact 4).  Written as
    ``y + (sum(p) + ...) * 0`` this costs four reductions and a dozen scalar kernels per AttnBlock and step, forward and backward; here it costs the zero fills."""

    @staticmethod
    def forward(ctx, y, *params):
        ctx.save_for_backward(*params)
        return y.view_as(y)

    @staticmethod
    def backward(ctx, grad_out):
        return (grad_out,) + tuple(torch.zeros_like(q) if need else None for q, need in zip(ctx.saved_tensors, ctx.needs_input_grad[1:]))


class _Ops:
    """Layer primitives over the parameter table; 3x3 convolutions and GroupNorms go to the HIP library when asked to and able to."""

    def __init__(self, params: P, hip: bool):
        self.p, self.hip = params, hip
        self.ss: Dict[str, torch.Tensor] = {}            # ResnetBlock name -> its (B, 2C) scale | shift rows (time_projections)

    def time_projections(self, t_act: torch.Tensor) -> None:
        """All ResnetBlock.mlp[1] Linears (Diffusion_arch.py:149-152; 20 of them, each a (B, 4 dim) x (4 dim, 2C) product of a few microseconds) as ONE
        Linear over the stacked weights -- what the sampling engine's ``tproj`` does: 1 + 2 GEMMs per step instead of 20 + 40 launch-bound ones.
        The stacked weight is a torch.cat of the parameters, so their gradients arrive as slices of one weight gradient."""
        names = [k[:-len(".mlp.1.weight")] for k, v in self.p.items() if k.endswith(".mlp.1.weight") and v.dim() == 2]
        if not names:
            return
        w = torch.cat([self.p[n + ".mlp.1.weight"] for n in names])
        b = torch.cat([self.p[n + ".mlp.1.bias"] for n in names])
        if self.hip and t_act.is_cuda:
            from . import train
            ss_all = train.linear(t_act, w, b)
        else:
            ss_all = F.linear(t_act, w, b)
        # one split (its backward is one cat of the 20 gradients; 20 slices would each zero-fill and accumulate a full-width gradient)
        for n, part in zip(names, ss_all.split([self.p[n + ".mlp.1.weight"].shape[0] for n in names], dim=1)):
            self.ss[n] = part

    def _left_library(self, name: str, op: str, why: str) -> None:
        """A layer of a .hip() network that runs on PyTorch's own kernel instead of the HIP library: said once per (layer, operator), on the
        `noisediff_amd.trainable` logger (the fallbacks by shape were silent), and counted in `FALLBACKS`."""
        if self.hip and (name, op) not in FALLBACKS:
            FALLBACKS[(name, op)] = why
            _log.warning("%s: %s stays on PyTorch (%s)", name, op, why)

    def conv(self, name: str, x, padding: int = 0, res: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``x``: a tensor, or a pair (x0, x1) standing for torch.cat((x0, x1), 1) -- the up path's skip concatenations, which the library's 3x3 and
        1x1 convolutions read as two sources (train.conv3x3_cat / conv1x1_cat) instead of a concatenated copy.  ``res``: conv(x) + res, the addition in the
        epilogue of the library's 1x1 kernel where that runs."""
        w, b = self.p[name + ".weight"], self.p.get(name + ".bias")
        if res is not None:
            if (self.hip and not isinstance(x, tuple) and x.is_cuda and w.shape[2:] == (1, 1) and padding == 0 and w.shape[0] % 4 == 0 and w.shape[1] % 4 == 0
                    and res.shape[1] == w.shape[0]):
                from . import train
                return train.conv1x1(x, w, b, res=res)
            return self.conv(name, x, padding) + res
        if isinstance(x, tuple):
            from . import train
            if self.hip and train.cat_sources_ok(*x, w.shape[0]):
                if w.shape[2:] == (3, 3) and padding == 1:
                    return train.conv3x3_cat(x[0], x[1], w, b)
                if w.shape[2:] == (1, 1) and padding == 0:
                    return train.conv1x1_cat(x[0], x[1], w, b)
            x = torch.cat(x, dim=1)
        if self.hip and x.is_cuda and w.shape[2:] == (3, 3) and padding == 1 and w.shape[0] % 8 == 0 and w.shape[1] % 8 == 0:
            from . import train
            return train.conv3x3(x, w, b)
        if self.hip and x.is_cuda and w.shape[2:] == (1, 1) and padding == 0 and w.shape[0] % 4 == 0 and w.shape[1] % 4 == 0:
            from . import train
            return train.conv1x1(x, w, b)
        if self.hip and x.is_cuda and tuple(w.shape[1:]) == (4, 7, 7) and padding == 3 and w.shape[0] % 4 == 0:
            from . import train
            return train.conv7x7_c4(x, w, b)                                    # the stem (init_conv / cond_init_conv)
        if self.hip and x.is_cuda and w.shape[2:] == (1, 1) and padding == 0 and w.shape[0] % 4 == 0 and w.shape[1] < 4:
            # pos_enc.weights (2 -> 8 channels, Diffusion_arch.py:328): as a 4-channel 1x1 convolution with two zero input channels -- the zero weight columns
            # receive gradients that the slice drops
            from . import train
            pad = 4 - w.shape[1]
            return train.conv1x1(F.pad(x, (0, 0, 0, 0, 0, pad)), F.pad(w, (0, 0, 0, 0, 0, pad)), b)
        if x.is_cuda:
            self._left_library(name, f"conv{w.shape[2]}x{w.shape[3]}", f"{w.shape[1]} -> {w.shape[0]} channels: the library's differentiable convolutions are 3x3 (channels % 8), 1x1 (cout % 4) and the 7x7 stem of a 4-channel image")
        return F.conv2d(x, w, b, padding=padding)

    def linear(self, name: str, x: torch.Tensor, res: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``res``: linear(x) + res, the addition in the GEMM's epilogue on the library."""
        w, b = self.p[name + ".weight"], self.p.get(name + ".bias")
        if self.hip and x.is_cuda and w.shape[0] % 4 == 0 and w.shape[1] % 4 == 0:
            from . import train
            return train.linear(x, w, b, res)
        if x.is_cuda:
            self._left_library(name, "linear", f"{w.shape[1]} -> {w.shape[0]}: channel counts must be multiples of 4")
        y = F.linear(x, w, b)
        return y if res is None else y + res

    def group_norm(self, name: str, x: torch.Tensor, groups: int) -> torch.Tensor:
        w, b = self.p[name + ".weight"], self.p[name + ".bias"]
        if self.hip and x.is_cuda:
            from . import train
            if train._group_norm_ok(x.shape[1], groups):                     # wider nets (dim > 128: C > 1024) stay on PyTorch's norm
                return train.group_norm(x, groups, w, b, 1e-5)
            self._left_library(name, "group_norm", f"C = {x.shape[1]}, {groups} groups: outside the library's norm (C <= 1024, C % 4 == 0)")
        return F.group_norm(x, groups, w, b, eps=1e-5)

    def layer_norm(self, name: str, x: torch.Tensor) -> torch.Tensor:
        w, b = self.p[name + ".weight"], self.p[name + ".bias"]
        if self.hip and x.is_cuda:
            from . import train
            if train._layer_norm_ok(x.shape[-1]):
                return train.layer_norm(x, w, b, 1e-5)
            self._left_library(name, "layer_norm", f"C = {x.shape[-1]}: the library's LayerNorm takes C = 64, 128 or a multiple of 256")
        return F.layer_norm(x, x.shape[-1:], w, b, eps=1e-5)

    # ---- composite layers ------------------------------------------------------------------------------------------
    def block(self, name: str, x, groups: int, ss=None, res=None):
        """Block: conv3x3 -> GroupNorm -> optional x (scale + 1) + shift -> SiLU   (Diffusion_arch.py:128-144); ``ss`` = scale | shift
        along dim 1: (B, 2C, 1, 1) from the time embedding or (B, 2C, H, W) per-pixel maps."""
        from . import train
        w = self.p[name + ".proj.weight"]
        pair = x if isinstance(x, tuple) else None                             # (x0, x1) = torch.cat((x0, x1), 1), read as two sources where the kernels can
        if pair is not None and not (self.hip and train.cat_sources_ok(*pair, w.shape[0]) and train._group_norm_ok(w.shape[0], groups)
                                     and (ss is None or ss.numel() == pair[0].shape[0] * 2 * w.shape[0])):
            x, pair = torch.cat(pair, dim=1), None
        x0 = pair[0] if pair is not None else x
        fused = (self.hip and x0.is_cuda and train._group_norm_ok(w.shape[0], groups)
                 and (ss is None or ss.numel() == x0.shape[0] * 2 * w.shape[0]))   # norm, per-sample modulation, SiLU (and the block's shortcut) as one operator
        if fused and w.shape[0] % 8 == 0 and w.shape[1] % 8 == 0:
            # ... fed by the convolution's statistics epilogue, as in the sampling engine: no pass of the norm's own over the conv output for its moments
            if pair is not None:
                x, cs = train.conv3x3_cat(pair[0], pair[1], w, self.p.get(name + ".proj.bias"), with_stats=True)
            else:
                x, cs = train.conv3x3_with_stats(x, w, self.p.get(name + ".proj.bias"))
            return train.group_norm_silu(x, groups, self.p[name + ".norm.weight"], self.p[name + ".norm.bias"], ss, 1e-5, res=res, conv_stats=cs)
        x = self.conv(name + ".proj", x, 1)
        if fused:
            return train.group_norm_silu(x, groups, self.p[name + ".norm.weight"], self.p[name + ".norm.bias"], ss, 1e-5, res=res)
        x = self.group_norm(name + ".norm", x, groups)
        if ss is not None and self.hip and x.is_cuda and x.shape[1] % 4 == 0 and ss.shape == (x.shape[0], 2 * x.shape[1]) + tuple(x.shape[2:]):
            x = train.modulate_silu(x, ss)                                      # per-pixel maps (ResnetBlock2): modulation + SiLU as one operator
        else:
            if ss is not None:
                scale, shift = ss.chunk(2, dim=1)
                x = x * (scale + 1) + shift
            x = F.silu(x)
        return x if res is None else x + res

    def resnet(self, name: str, x, emb, groups: int, per_pixel: bool = False):
        """ResnetBlock (per-sample scale / shift from the time embedding, :146-170) and ResnetBlock2 (per-pixel maps from the
        position embedding, :173-196); the shortcut is a 1x1 conv iff the channel count changes."""
        ss = None
        if emb is not None:      # ``emb`` arrives ACTIVATED: every block's mlp starts with the same SiLU of the same embedding (:149,176) -- computed once in _forward
            if per_pixel:
                ss = self.conv(name + ".mlp.1", emb)
            elif name in self.ss:                                # the block's rows of the one stacked projection (time_projections)
                ss = self.ss[name][:, :, None, None]
            else:
                ss = self.linear(name + ".mlp.1", emb)[:, :, None, None]
        if isinstance(x, tuple) and name + ".res_conv.weight" not in self.p:
            x = torch.cat(x, dim=1)                                            # an identity shortcut needs the tensor itself
        if isinstance(x, tuple) and self.hip:
            from . import train
            w = self.p[name + ".res_conv.weight"]
            if train.cat_sources_ok(*x, w.shape[0]):               # (what conv() asks before it reads the pair as two sources)
                # the shortcut hands the pair on to block1: that convolution's data gradient then joins the shortcut's in one GEMM instead of two additions
                x0, x1, res = train.conv1x1_shortcut_cat(x[0], x[1], w, self.p.get(name + ".res_conv.bias"))
                return self.block(name + ".block2", self.block(name + ".block1", (x0, x1), groups, ss), groups, res=res)
        res = self.conv(name + ".res_conv", x) if name + ".res_conv.weight" in self.p else x
        return self.block(name + ".block2", self.block(name + ".block1", x, groups, ss), groups, res=res)      # block2(...) + res, the add inside block2's fused tail

    def mlp(self, name: str, x):
        """Mlp: conv1x1 -> GELU -> conv1x1   (:340-356)."""
        return self.conv(name + ".fc2", F.gelu(self.conv(name + ".fc1", x)))

    def cross_attention(self, name: str, x, ctx):
        """CrossAttention over the ISO context (:361-402).  The context is ONE token (iso_embed row, :591), so the softmax runs over a
        single key: its weights are exactly 1 and its derivative exactly 0 -- the output is to_out(to_v(ctx)) for every query token,
        bit for bit, and to_q / to_k (and the LayerNorm that feeds only them) receive exactly-zero gradients in the reference too
        (the sampling path uses the same identity).  The general form stays for longer contexts."""
        b, n, _ = x.shape
        if ctx.shape[1] == 1:
            out = self.linear(name + ".to_out.0", self.linear(name + ".to_v", ctx))              # (b, 1, C): broadcasts over the n tokens
            return _ZeroGrads.apply(out, self.p[name + ".to_q.weight"], self.p[name + ".to_k.weight"])   # zero gradients, as autograd gives them there
        q, k, v = self.linear(name + ".to_q", x), self.linear(name + ".to_k", ctx), self.linear(name + ".to_v", ctx)
        d = q.shape[-1] // HEADS
        heads = lambda t: t.reshape(b, t.shape[1], HEADS, d).transpose(1, 2)               # b h n d
        q, k, v = heads(q), heads(k), heads(v)
        attn = (q @ k.transpose(-1, -2) * d ** -0.5).softmax(dim=-1)
        out = (attn @ v).transpose(1, 2).reshape(b, n, HEADS * d)
        return self.linear(name + ".to_out.0", out)

    def attn_block(self, name: str, x, ctx):
        """AttnBlock: tokens; x += attn(LN1 x, ctx); x += FF(LN2 x); proj_out + input   (:425-443)."""
        b, c, h, w = x.shape
        t = x.flatten(2).transpose(1, 2)
        if ctx.shape[1] == 1:                                                # norm1 feeds only the (dead) queries: skipped, its gradient is zero
            vec = _ZeroGrads.apply(self.cross_attention(name + ".attn", t, ctx), self.p[name + ".norm1.weight"], self.p[name + ".norm1.bias"])   # (b, 1, C): the same vector for every token
            if self.hip and t.is_cuda:
                from . import train
                t = train.broadcast_add(t, vec)                                     # its gradient: a token sum on the library (fixed order)
            else:
                t = vec + t
        else:
            t = self.cross_attention(name + ".attn", self.layer_norm(name + ".norm1", t), ctx) + t
        t = self.linear(name + ".ff.net.2", F.gelu(self.linear(name + ".ff.net.0.0", self.layer_norm(name + ".norm2", t))), res=t)
        return self.conv(name + ".proj_out", t.transpose(1, 2).reshape(b, c, h, w), res=x)


    # ---- the attention modules the reference defines next to the network (Diffusion_arch.py:84-90,198-266): BASELINE config 4's mid-block Attention and
    #      upstream's per-stage LinearAttention / Attention wiring.  Plain differentiable PyTorch around the library's 1x1 convolutions.
    def rms_norm(self, name: str, x: torch.Tensor) -> torch.Tensor:
        """RMSNorm (:84-90): F.normalize over the channels * g * sqrt(C)."""
        return F.normalize(x, dim=1) * self.p[name + ".g"] * (x.shape[1] ** 0.5)

    def attention(self, name: str, x: torch.Tensor) -> torch.Tensor:
        """Attention (:237-266) with Attend's explicit path (models/attend.py:101-116): RMSNorm -> to_qkv -> softmax(q k^T / sqrt(d)) v -> to_out."""
        b, c, h, w = x.shape
        q, k, v = (t.reshape(b, HEADS, -1, h * w).transpose(-1, -2) for t in self.conv(name + ".to_qkv", self.rms_norm(name + ".norm", x)).chunk(3, dim=1))   # b h (xy) d
        out = F.scaled_dot_product_attention(q, k, v)                            # scale d^-1/2, no mask, no dropout
        return self.conv(name + ".to_out", out.transpose(-1, -2).reshape(b, -1, h, w))

    def linear_attention(self, name: str, x: torch.Tensor) -> torch.Tensor:
        """LinearAttention (:198-235): q softmax over d (scaled), k softmax over the pixels, context = k v^T, out = context^T q -> to_out.0 -> RMSNorm."""
        b, c, h, w = x.shape
        q, k, v = (t.reshape(b, HEADS, -1, h * w) for t in self.conv(name + ".to_qkv", self.rms_norm(name + ".norm", x)).chunk(3, dim=1))    # b h d (xy)
        q = q.softmax(dim=-2) * (q.shape[2] ** -0.5)
        k = k.softmax(dim=-1)
        ctx = torch.einsum("bhdn,bhen->bhde", k, v)
        out = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(b, -1, h, w)
        return self.rms_norm(name + ".to_out.1", self.conv(name + ".to_out.0", out))

    def stage_attention(self, name: str, kind, x: torch.Tensor) -> torch.Tensor:
        if not kind:
            return x
        return (self.attention(name, x) if kind == "full" else self.linear_attention(name, x)) + x


def _forward(o: _Ops, x: torch.Tensor, time: torch.Tensor, condition, arch: str = "NoiseDiffNet", mid_attn: bool = False, stage_attn=None) -> torch.Tensor:
    """NoiseDiffNet.forward (Diffusion_arch.py:577-646: shot-noise branch on cat(clean, x) + the U-Net's read-noise branch) and the forward of the
    ``UNet_PosEmbV2*`` ablation nets (others_arch.py:483-537, :655-707, :919-985: no shot branch; the clean image through cond_init_conv -> cond_res_block1 ->
    cond_concat_conv; ``_NoPosition`` without the position inputs, ``_CameraCond`` with the ISO AttnBlocks).  ``mid_attn``: ``x = Attention(x) + x`` between
    the mid blocks (BASELINE config 4); ``stage_attn``: four entries 'linear' / 'full' / None, upstream's per-stage wiring (modules down_attns.{i} / up_attns.{i})."""
    from .spec import arch_traits
    tr = arch_traits(arch)
    p = o.p
    dim = p["init_conv.weight"].shape[0]
    if x.shape[-1] % 8 or x.shape[-2] % 8:
        raise ValueError(f"{arch} needs image sides that are multiples of 8, got {tuple(x.shape[-2:])}")
    clean = condition["clean_img"] if isinstance(condition, dict) else condition           # (_NoPosition takes the clean image itself, others_arch.py:658)
    kinds = tuple(stage_attn) if stage_attn else (None,) * 4
    # condition embeddings: learned sinusoidal position features -> Mlp; ISO table row as a one-token context; time MLP
    pos = None
    if tr.position:
        w = o.conv("pos_enc.weights", condition["position"])
        pos = o.mlp("pos_mlp", torch.cat((w, (2 * math.pi * w).sin(), (2 * math.pi * w).cos()), dim=1))
    iso = None
    if tr.iso_attn:
        iso = F.embedding(condition["iso_ratio_idx"].long().to(x.device), p["iso_embed.weight"])[:, None]      # (the trainer keeps the index on the CPU, trainer_diffusion.py:135)
    half = dim // 2
    freqs = torch.exp(torch.arange(half, device=x.device) * -(math.log(10000.0) / (half - 1)))
    ang = time[:, None] * freqs[None]
    t = o.linear("time_mlp.3", F.gelu(o.linear("time_mlp.1", torch.cat((ang.sin(), ang.cos()), dim=-1))))

    # ResnetBlock.mlp / ResnetBlock2.mlp = Sequential(SiLU, Linear / Conv2d) of the SAME embedding in all 20 (2) blocks: one activation each, not 20
    # forward + 20 backward + 19 gradient accumulations of launch-bound (4, 256) kernels (and two full-resolution SiLUs of the position embedding)
    t = F.silu(t)
    if pos is not None:
        pos = F.silu(pos)
    o.time_projections(t)
    shot = None
    if tr.shot_branch:
        s0 = o.mlp("shot_mlp1", torch.cat((clean, x), dim=1))
        s = o.mlp("shot_mlp2", o.attn_block("shot_attn", s0, iso))
        shot = o.mlp("shot_mlp3", o.resnet("shot_time", s, t, POS_GROUPS) + s0)

    def stem7(name: str, v: torch.Tensor) -> torch.Tensor:
        v = o.conv(name, v, 3)
        if o.hip and v.is_cuda:                 # MIOpen answers the 4-channel NCHW input in NCHW: one conversion here instead of one in every consumer of the stem
            v = v.contiguous(memory_format=torch.channels_last)   # (pos_block1's conv, its shortcut, the final concat and both of that concat's readers)
        return v

    x = stem7("init_conv", x)
    stem = x
    if tr.cond_branch:                          # x = cond_concat_conv(cat[init_conv(x), cond_res_block1(cond_init_conv(clean))])   others_arch.py:491-498
        clean_emb = o.resnet("cond_res_block1", stem7("cond_init_conv", clean), None, GROUPS)
        x = o.conv("cond_concat_conv", torch.cat((x, clean_emb), dim=1), 1)
    pos_block = (lambda n, v: o.resnet(n, v, pos, POS_GROUPS, per_pixel=True)) if tr.position else (lambda n, v: o.resnet(n, v, None, POS_GROUPS))
    x = pos_block("pos_block1", x)
    rs = 3 if tr.iso_attn else 2                # index of a stage's resampling layer
    skips: List[torch.Tensor] = []
    for i in range(4):
        n = f"downs.{i}"
        x = o.resnet(n + ".0", x, t, GROUPS); skips.append(x)
        x = o.stage_attention(f"down_attns.{i}", kinds[i], o.resnet(n + ".1", x, t, GROUPS)); skips.append(x)
        if tr.iso_attn:
            x = o.attn_block(n + ".2", x, iso)
        x = o.conv(f"{n}.{rs}", x, 1) if i == 3 else o.conv(f"{n}.{rs}.1", F.pixel_unshuffle(x, 2))
    x = o.resnet("mid_block1", x, t, GROUPS)
    if mid_attn:
        x = o.attention("mid_attn", x) + x
    x = o.resnet("mid_block2", x, t, GROUPS)
    for i in range(4):
        n = f"ups.{i}"
        x = o.resnet(n + ".0", (x, skips.pop()), t, GROUPS)                    # (a pair = the concatenation, read as two sources)
        x = o.stage_attention(f"up_attns.{i}", kinds[3 - i], o.resnet(n + ".1", (x, skips.pop()), t, GROUPS))
        if tr.iso_attn:
            x = o.attn_block(n + ".2", x, iso)
        x = o.conv(f"{n}.{rs}", x, 1) if i == 3 else o.conv(f"{n}.{rs}.1", F.interpolate(x, scale_factor=2, mode="nearest"), 1)
    x = pos_block("pos_block2", x)
    x = o.resnet("final_res_block", (x, stem), t, GROUPS)
    out = o.conv("final_conv", x)
    return out if shot is None else shot + out


class TrainableNoiseDiffNet(nn.Module):
    """``TrainableNoiseDiffNet(args)``: args.dim (default 64), the other fields of the reference's argument object are accepted and
    must describe the configuration this package implements (4 input channels, no self-conditioning).  Optional: ``args.arch`` (one of the
    ``UNet_PosEmbV2*`` ablation nets of others_arch.py instead of NoiseDiffNet), ``args.mid_attn`` (BASELINE config 4's mid-block Attention),
    ``args.stage_attn`` (upstream's per-stage attention wiring: True or four entries) -- the extensions of noisediff_amd.NoiseDiffNet, under its names."""
    channels = out_dim = 4
    self_condition = False
    random_or_learned_sinusoidal_cond = False

    def __init__(self, args=None, seed: int = 0):
        super().__init__()
        from .spec import arch_param_spec, attention_param_spec, normalize_stage_attn, stage_attention_param_spec
        dim = int(getattr(args, "dim", 64))
        if getattr(args, "self_condition", False) or int(getattr(args, "inp_dim", 4)) != 4 or int(getattr(args, "cond_dim", 4)) != 4:
            raise ValueError("TrainableNoiseDiffNet implements the reference's configuration: inp_dim = cond_dim = 4, self_condition = False")
        self.dim = dim
        self.arch = str(getattr(args, "arch", "NoiseDiffNet"))
        self.has_mid_attn = bool(getattr(args, "mid_attn", False))
        self.stage_attn = normalize_stage_attn(getattr(args, "stage_attn", None))
        if self.arch != "NoiseDiffNet" and (self.has_mid_attn or self.stage_attn):
            raise ValueError("mid_attn / stage_attn extend NoiseDiffNet only")
        self._hip = False
        spec = list(arch_param_spec(self.arch, dim, 4))
        if self.has_mid_attn:
            spec += attention_param_spec("mid_attn", 8 * dim)
        if self.stage_attn:
            spec += stage_attention_param_spec(dim, self.stage_attn)
        for name, value in synth.make_state_dict(spec, seed).items():      # PyTorch's default-init statistics
            parts, m = name.split("."), self
            for part in parts[:-1]:
                if part not in m._modules:
                    m.add_module(part, nn.Module())
                m = m._modules[part]
            m.register_parameter(parts[-1], nn.Parameter(value))

    def hip(self, on: bool = True) -> "TrainableNoiseDiffNet":
        """3x3 convolutions and GroupNorms forward and backward, Linear / 1x1 weight gradients on libnoisediff_hip (CUDA tensors only; raises without the library)."""
        if on:
            from . import _lib
            _lib.load()
        self._hip = bool(on)
        return self

    def forward(self, x: torch.Tensor, time: torch.Tensor, condition: Dict[str, torch.Tensor], x_self_cond: Optional[torch.Tensor] = None):
        if x_self_cond is not None:
            raise ValueError("self-conditioning is not part of this configuration")
        cache = self.__dict__.get("_nd_param_table")                        # (name -> Parameter, [(owner's _parameters, key, Parameter)]): named_parameters() costs 1.5 ms per step
        if cache is None or cache[2] != id(self) or not all(d.get(k) is q for d, k, q in cache[1]):     # a replaced Parameter rebuilds it; conversions (.to, .float) keep the
            # objects; an nn.DataParallel replica (a shallow copy of this module's __dict__ with its own broadcast tensors) must not inherit the owner's table
            table, checks = {}, []
            for prefix, m in self.named_modules():
                for k, q in m._parameters.items():
                    if q is not None:
                        table[f"{prefix}.{k}" if prefix else k] = q
                        checks.append((m._parameters, k, q))
            cache = self.__dict__["_nd_param_table"] = (table, checks, id(self))
        table = cache[0]
        return _forward(_Ops(table, self._hip), x, time, condition, arch=self.arch, mid_attn=self.has_mid_attn, stage_attn=self.stage_attn)
