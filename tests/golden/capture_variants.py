#!/usr/bin/env python3
"""Golden vectors for SURVEY 8f-3: the UNet_PosEmbV2* ablation nets of the REAL reference (build container only).

    python tests/golden/capture_variants.py      # writes tests/golden/variants.npz

Weights, conditions and inputs come from noisediff_amd.synth (hash streams), so the fixture holds outputs only:
the whole-net forward at d=16, 32x32, B=2 and t in {3, 500, 999}, strided taps of the intermediates, and one
8-step DDIM run of the reference's GaussianDiffusion around each net (reference noise calls patched to the named
streams exactly as capture_golden.py does)."""
import os, sys, types
from types import SimpleNamespace
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
from noisediff_amd import synth
from noisediff_amd.spec import posemb_unet_param_spec
from capture_golden import PatchedNoise, import_reference, sub

ddp, _ = import_reference()
import models.archs.others_arch as oa

ARCHS = ("UNet_PosEmbV2", "UNet_PosEmbV2_NoPosition", "UNet_PosEmbV2_CameraCond")
DIM, B, S = 16, 2, 32
TAPS = {"clean_emb": "cond_res_block1", "cond_concat": "cond_concat_conv", "pos_block1": "pos_block1",
        "mid": "mid_block2", "pos_block2": "pos_block2"}
out = {}
with torch.no_grad():
    for arch in ARCHS:
        args = SimpleNamespace(dim=DIM, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False)
        net = getattr(oa, arch)(args).eval()
        spec = posemb_unet_param_spec(arch, DIM)
        assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == [(p.name, p.shape) for p in spec]
        net.load_state_dict(synth.make_state_dict(spec, 0), strict=True)
        cond = synth.make_condition(B, S, seed=1)
        ref_cond = cond["clean_img"] if arch == "UNet_PosEmbV2_NoPosition" else cond      # others_arch.py:658
        x = synth.make_noise(3, f"var.{arch}.x", B, 4, S)
        got = {}
        hooks = [getattr(net, mod).register_forward_hook(lambda m, i, o, k=key: got.__setitem__(k, o)) for key, mod in TAPS.items()]
        for t in (3, 500, 999):
            y = net(x, torch.full((B,), t, dtype=torch.long), ref_cond)
            out[f"{arch}.out.t{t}"] = y.numpy()
        for h in hooks:
            h.remove()
        for k, v in got.items():                      # taps of the last forward (t = 999)
            out[f"{arch}.tap.{k}"] = sub(v)
        gd = ddp.GaussianDiffusion(torch.nn.DataParallel(net), image_size=S, timesteps=1000, sampling_timesteps=8,
                                   beta_schedule="sigmoid2", objective="pred_v")
        with PatchedNoise(5, B, 4, S):
            out[f"{arch}.ddim8"] = gd.sample(batch_size=B, condition=ref_cond).numpy()
np.savez_compressed(os.path.join(HERE, "variants.npz"), **out)
print({k: v.shape for k, v in out.items()})
