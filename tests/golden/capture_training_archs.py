#!/usr/bin/env python3
"""Generate tests/golden/training_archs.npz: the reference's training loss and parameter gradients for every network this package trains besides
the plain NoiseDiffNet of capture_training.py (build container only).

    python tests/golden/capture_training_archs.py

Cases (d=16, 32x32, B=2, the 'train.*' streams of capture_training.py, p_losses with objective pred_v -- models/denoising_diffusion_pytorch.py:481-531):
  * UNet_PosEmbV2, UNet_PosEmbV2_NoPosition, UNet_PosEmbV2_CameraCond  (models/archs/others_arch.py:364-985), constructed by the reference;
  * NoiseDiffNet + the mid-block Attention of BASELINE config 4 (``x = attn(x) + x`` between mid_block1 and mid_block2: the reference's own
    ``Attention`` class attached with a forward hook, as capture_golden.py does for the sampling goldens);
  * NoiseDiffNet + upstream's per-stage LinearAttention / Attention wiring (forward hooks, as capture_stage_attn.py).
The attention modules' parameters take part in the backward pass; their gradients are recorded under this package's names (``mid_attn.*``,
``down_attns.{i}.*``, ``up_attns.{i}.*``).  The fixture holds outputs only: the loss, the squared norm of the whole gradient and strided samples of a few
parameter gradients per case.
"""
from __future__ import annotations

import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from capture_golden import import_reference, ref_net, sub, synth  # noqa: E402
from capture_training import B, DIM, H, T, inputs  # noqa: E402
from noisediff_amd.spec import (STAGE_ATTN_REFERENCE, attention_param_spec, noisediff_param_spec, posemb_unet_param_spec,  # noqa: E402
                                stage_attention_param_spec, stage_dims)

VARIANT_GRADS = ["final_conv.weight", "cond_init_conv.weight", "cond_concat_conv.weight", "downs.0.0.block1.proj.weight", "time_mlp.1.weight",
                 "mid_block1.block2.norm.weight", "pos_block1.block1.proj.weight"]
NET_GRADS = ["final_conv.weight", "downs.3.1.block2.proj.weight", "mid_block2.block1.proj.weight", "time_mlp.1.weight"]


def record(out, case, loss, named):
    out[f"{case}.loss"] = np.float64(loss.item())
    out[f"{case}.grad_sq_norm"] = np.float64(sum(float((p.grad.double() ** 2).sum()) for p in named.values() if p.grad is not None))
    out[f"{case}.n_params_with_grad"] = np.int64(sum(1 for p in named.values() if p.grad is not None))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ddp, arch = import_reference()
    import models.archs.others_arch as oa
    out = {}
    x0, noise, t, cond = inputs()
    args = SimpleNamespace(dim=DIM, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False)

    # ---- the ablation nets
    for name in ("UNet_PosEmbV2", "UNet_PosEmbV2_NoPosition", "UNet_PosEmbV2_CameraCond"):
        net = getattr(oa, name)(args).train()
        net.load_state_dict(synth.make_state_dict(posemb_unet_param_spec(name, DIM), 0), strict=True)
        gd = ddp.GaussianDiffusion(torch.nn.DataParallel(net), image_size=H, timesteps=T, beta_schedule="sigmoid2", objective="pred_v")
        ref_cond = cond["clean_img"] if name == "UNet_PosEmbV2_NoPosition" else cond      # others_arch.py:658
        loss = gd.p_losses(x0, t, ref_cond, noise=noise.clone())
        loss.backward()
        named = dict(net.named_parameters())
        record(out, name, loss, named)
        for k in VARIANT_GRADS + (["downs.1.2.ff.net.2.weight", "iso_embed.weight"] if name.endswith("CameraCond") else []):
            if k in named and named[k].grad is not None:
                out[f"{name}.grad.{k}"] = sub(named[k].grad, 2048)

    # ---- NoiseDiffNet + mid-block Attention (BASELINE config 4)
    net = ref_net(arch, DIM).train()
    C = 8 * DIM
    att = arch.Attention(C, heads=4, dim_head=32, flash=False).train()
    att.load_state_dict({k[len("mid_attn."):]: v for k, v in synth.make_state_dict(attention_param_spec("mid_attn", C), 0).items()}, strict=True)
    net.mid_block1.register_forward_hook(lambda _m, _i, o: att(o) + o)
    gd = ddp.GaussianDiffusion(torch.nn.DataParallel(net), image_size=H, timesteps=T, beta_schedule="sigmoid2", objective="pred_v")
    loss = gd.p_losses(x0, t, cond, noise=noise.clone())
    loss.backward()
    named = {**dict(net.named_parameters()), **{"mid_attn." + k: v for k, v in att.named_parameters()}}
    record(out, "mid_attn", loss, named)
    for k in NET_GRADS + ["mid_attn.to_qkv.weight", "mid_attn.to_out.weight", "mid_attn.norm.g"]:
        out[f"mid_attn.grad.{k}"] = sub(named[k].grad, 2048)

    # ---- NoiseDiffNet + per-stage LinearAttention x 3 / Attention (the reference's own full_attn tuple)
    net = ref_net(arch, DIM).train()
    kinds = STAGE_ATTN_REFERENCE
    sda = synth.make_state_dict(stage_attention_param_spec(DIM, kinds), 0)
    dims = stage_dims(DIM)
    mods = {}

    def attach(block, prefix, kind, width):
        m = (arch.Attention(width, heads=4, dim_head=32, flash=False) if kind == "full" else arch.LinearAttention(width, heads=4, dim_head=32)).train()
        m.load_state_dict({k[len(prefix) + 1:]: v for k, v in sda.items() if k.startswith(prefix + ".")}, strict=True)
        mods[prefix] = m
        block.register_forward_hook(lambda _m, _i, o, m=m: m(o) + o)

    for i, kind in enumerate(kinds):
        attach(net.downs[i][1], f"down_attns.{i}", kind, dims[i][0])
        attach(net.ups[i][1], f"up_attns.{i}", kinds[3 - i], dims[3 - i][1])
    gd = ddp.GaussianDiffusion(torch.nn.DataParallel(net), image_size=H, timesteps=T, beta_schedule="sigmoid2", objective="pred_v")
    loss = gd.p_losses(x0, t, cond, noise=noise.clone())
    loss.backward()
    named = dict(net.named_parameters())
    for prefix, m in mods.items():
        named.update({f"{prefix}.{k}": v for k, v in m.named_parameters()})
    record(out, "stage_attn", loss, named)
    for k in NET_GRADS + ["down_attns.0.to_qkv.weight", "down_attns.3.to_out.weight", "up_attns.0.to_qkv.weight", "up_attns.3.to_out.1.g"]:
        out[f"stage_attn.grad.{k}"] = sub(named[k].grad, 2048)

    np.savez_compressed(os.path.join(HERE, "training_archs.npz"), **out)
    for k, v in out.items():
        print(k, v if np.ndim(v) == 0 else np.shape(v))


if __name__ == "__main__":
    main()
