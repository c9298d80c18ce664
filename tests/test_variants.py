"""the UNet_PosEmbV2* ablation nets (models/archs/others_arch.py:364-985).

CPU part: spec and oracle against fixtures captured from the real reference (tests/golden/capture_variants.py).
GPU part (-m gpu): the same nets on the HIP kernels, through the reference's plug-in interface."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from noisediff_amd import synth
from noisediff_amd.spec import ARCHS, arch_param_spec, arch_traits
from oracle import noisediff_oracle as O
from util import noise_fn, rel_err, sub

VARIANTS = [a for a in ARCHS if a != "NoiseDiffNet"]
DIM, B, S = 16, 2, 32
N_KEYS = {"UNet_PosEmbV2": 270, "UNet_PosEmbV2_NoPosition": 260, "UNet_PosEmbV2_CameraCond": 391}   # reference state dicts


def _inputs(arch):
    sd = synth.make_state_dict(arch_param_spec(arch, DIM), 0)
    cond = synth.make_condition(B, S, seed=1)
    x = synth.make_noise(3, f"var.{arch}.x", B, 4, S)
    return sd, cond, x


@pytest.mark.parametrize("arch", VARIANTS)
def test_variant_spec_and_oracle_match_reference(golden, arch):
    spec = arch_param_spec(arch, DIM)
    assert len(spec) == N_KEYS[arch]       # names and shapes were compared with the reference's state dict at capture time
    tr = arch_traits(arch)
    assert tr.cond_branch and not tr.shot_branch
    sd, cond, x = _inputs(arch)
    with torch.no_grad():
        for t in (3, 500, 999):
            taps = {}
            y = O.posemb_unet_forward(sd, arch, x, torch.full((B,), t, dtype=torch.long), cond, taps=taps)
            assert rel_err(y.numpy(), golden("variants", f"{arch}.out.t{t}")) < 2e-5, t
        for k in ("clean_emb", "cond_concat", "pos_block1", "mid", "pos_block2"):
            assert rel_err(sub(taps[k]), golden("variants", f"{arch}.tap.{k}")) < 2e-5, k
        if arch == "UNet_PosEmbV2_NoPosition":      # the reference passes the bare clean image here (others_arch.py:658)
            y2 = O.posemb_unet_forward(sd, arch, x, torch.full((B,), 999, dtype=torch.long), cond["clean_img"])
            assert torch.equal(y, y2)


@pytest.mark.parametrize("arch", VARIANTS)
def test_variant_ddim8_oracle_matches_reference(golden, arch):
    sd, cond, _ = _inputs(arch)
    buf = O.schedule_buffers("sigmoid2", 1000, "pred_v")
    with torch.no_grad():
        res = O.ddim_sample(lambda v, t: O.posemb_unet_forward(sd, arch, v, t, cond), buf, "pred_v",
                            synth.make_noise(5, "x_T", B, 4, S), noise_fn(5, B, 4, S), 8)
    assert rel_err(res.numpy(), golden("variants", f"{arch}.ddim8")) < 1e-4


# --------------------------------------------------------------------------------------------- HIP (MI355X)

@pytest.mark.gpu
@pytest.mark.parametrize("arch", VARIANTS)
def test_variant_hip_forward_and_sampler(golden, arch):
    import noisediff_amd as nd
    dev = torch.device("cuda", 0)
    sd, cond, x = _inputs(arch)
    args = SimpleNamespace(dim=DIM, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False)
    net = getattr(nd, arch)(args)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    cond_d = {k: v.to(dev) for k, v in cond.items()}
    ref_cond = cond_d["clean_img"] if arch == "UNet_PosEmbV2_NoPosition" else cond_d
    with torch.inference_mode():
        for t in (3, 500, 999):
            y = net(x.to(dev), torch.full((B,), t, dtype=torch.long, device=dev), ref_cond)
            assert rel_err(y.cpu().numpy(), golden("variants", f"{arch}.out.t{t}")) < 2e-4, t
        # taps of the last forward through the debug plan
        plan = net.hip_engine(dev).plan(B, S, S, debug=True)
        plan.set_condition(ref_cond)
        plan.forward(x.to(dev), torch.full((B,), 999, dtype=torch.long, device=dev))
        for k in ("clean_emb", "cond_concat", "pos_block1", "mid", "pos_block2"):
            got = plan.taps[k].permute(0, 3, 1, 2).contiguous().cpu()
            assert rel_err(sub(got), golden("variants", f"{arch}.tap.{k}")) < 2e-4, k
        gd = nd.GaussianDiffusion(torch.nn.DataParallel(net), image_size=S, timesteps=1000, sampling_timesteps=8,
                                  beta_schedule="sigmoid2", objective="pred_v").to(dev)
        steps = torch.stack([synth.make_noise(5, f"noise.{i}", B, 4, S) for i in range(7)])
        res = gd.sample(batch_size=B, condition=ref_cond, noise={"x_T": synth.make_noise(5, "x_T", B, 4, S), "steps": steps})
    assert rel_err(res.cpu().numpy(), golden("variants", f"{arch}.ddim8")) < 1e-3      # north-star tolerance


@pytest.mark.gpu
@pytest.mark.parametrize("arch", VARIANTS)
def test_variant_hip_forward_at_bench_width_matches_oracle(arch):
    """d=64 (the benchmark width: Winograd convs, fused chains and the pipelined pointwise GEMM all active), 64x64, B=2,
    against the oracle (itself pinned by the d=16 reference goldens above)."""
    import noisediff_amd as nd
    dev = torch.device("cuda", 0)
    dim, b, size = 64, 2, 64
    sd = synth.make_state_dict(arch_param_spec(arch, dim), 0)
    cond = synth.make_condition(b, size, seed=1)
    x = synth.make_noise(3, f"var64.{arch}.x", b, 4, size)
    t = torch.tensor([17, 803], dtype=torch.long)
    net = getattr(nd, arch)(SimpleNamespace(dim=dim, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False))
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    with torch.inference_mode():
        y = net(x.to(dev), t.to(dev), {k: v.to(dev) for k, v in cond.items()}).cpu()
    with torch.no_grad():
        ref = O.posemb_unet_forward(sd, arch, x, t, cond)
    assert rel_err(y.numpy(), ref.numpy()) < 2e-4
