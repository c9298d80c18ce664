#!/usr/bin/env python3
"""Golden vectors for SURVEY 8f-3, second half: upstream's per-stage attention wiring on the REAL reference (build container only).

    python tests/golden/capture_stage_attn.py      # writes tests/golden/stage_attn.npz

The reference defines LinearAttention / Attention (models/archs/Diffusion_arch.py:198-266), computes ``full_attn = (False, False, False,
True)`` and ``FullAttention`` per stage (:467-468,509-518) and never puts them into its ModuleLists.  Upstream applies
``attn_klass(dim)`` as ``x = attn(x) + x`` behind a stage's second ResnetBlock, in front of the skip.  This script reproduces exactly that
on the reference network with forward hooks -- ``downs[i][1]`` / ``ups[i][1]`` return ``attn(out) + out`` with the reference's own classes,
as capture_golden.py does for the mid-block Attention of BASELINE config 4 -- and records the whole-net forward at d=16, 64x64, B=2 for
three timesteps, strided taps of the intermediates and one 6-step DDIM run of the reference's GaussianDiffusion.  Weights, conditions,
inputs and noise come from noisediff_amd.synth (hash streams): the fixture holds outputs only."""
import os, sys
from types import SimpleNamespace
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
from noisediff_amd import synth
from noisediff_amd.spec import STAGE_ATTN_REFERENCE, noisediff_param_spec, stage_attention_param_spec, stage_dims
from capture_golden import PatchedNoise, import_reference, sub

ddp, arch = import_reference()
DIM, B, S = 16, 2, 64
KINDS = STAGE_ATTN_REFERENCE                     # the reference's own full_attn tuple: LinearAttention x 3, full Attention at the last stage


def wired_net():
    net = arch.NoiseDiffNet(SimpleNamespace(dim=DIM, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False)).eval()
    net.load_state_dict(synth.make_state_dict(noisediff_param_spec(DIM), 0), strict=True)
    sda = synth.make_state_dict(stage_attention_param_spec(DIM, KINDS), 0)
    dims = stage_dims(DIM)
    hooks = []

    def attach(block, prefix, kind, width):
        m = (arch.Attention(width, heads=4, dim_head=32, flash=False) if kind == "full" else arch.LinearAttention(width, heads=4, dim_head=32)).eval()
        m.load_state_dict({k[len(prefix) + 1:]: v for k, v in sda.items() if k.startswith(prefix + ".")}, strict=True)   # proves the spec's names / shapes
        hooks.append(block.register_forward_hook(lambda _m, _i, o, m=m: m(o) + o))

    for i, kind in enumerate(KINDS):
        attach(net.downs[i][1], f"down_attns.{i}", kind, dims[i][0])
        attach(net.ups[i][1], f"up_attns.{i}", KINDS[3 - i], dims[3 - i][1])
    return net, hooks


out = {}
with torch.no_grad():
    net, _ = wired_net()
    cond = synth.make_condition(B, S, seed=1)
    x = synth.make_noise(4, "sa.x", B, 4, S)
    taps = {}
    names = {"down0": net.downs[0][3], "down3": net.downs[3][3], "mid": net.mid_block2, "up0": net.ups[0][3], "up3": net.ups[3][3]}
    th = [m.register_forward_hook(lambda _m, _i, o, k=k: taps.__setitem__(k, o)) for k, m in names.items()]
    for t in (3, 500, 999):
        y = net(x, torch.full((B,), t, dtype=torch.long), cond)
        out[f"sa.fwd.t{t}"] = y.numpy()
        if t == 500:
            for k, v in taps.items():
                out[f"sa.tap.{k}"] = sub(v, 4096)
    for h in th:
        h.remove()
    # the sampler around the wired net: 6-step DDIM, eta 0.5 (noise path on), reference noise calls patched to the named streams
    gd = ddp.GaussianDiffusion(torch.nn.DataParallel(net), image_size=S, timesteps=1000, sampling_timesteps=6, beta_schedule="sigmoid2",
                               objective="pred_v", ddim_sampling_eta=0.5)
    with PatchedNoise(2, B, 4, S):
        out["sa.samp.ddim6"] = gd.sample(batch_size=B, condition=cond).numpy()
np.savez_compressed(os.path.join(HERE, "stage_attn.npz"), **out)
print({k: v.shape for k, v in out.items()})
