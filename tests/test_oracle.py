"""Pin the CPU oracle against fixtures captured from the real reference."""
import numpy as np
import pytest
import torch

from noisediff_amd import synth
from noisediff_amd.spec import noisediff_param_spec
from oracle import noisediff_oracle as O
from util import noise_fn, rel_err, state_dict, sub

TOL = 2e-5


def test_state_dict_keys_match_reference(meta):
    for dim in (16, 32, 48, 64):
        ours = [[p.name, list(p.shape)] for p in noisediff_param_spec(dim)]
        assert ours == meta[f"state_dict.d{dim}"]
        assert len(ours) == 416


@pytest.mark.parametrize("sched", ["linear", "cosine", "sigmoid1", "sigmoid2", "sigmoid3"])
@pytest.mark.parametrize("T", [1000, 20])
def test_schedule_buffers(golden, sched, T):
    names = [str(n) for n in golden("schedules", "sched.names")]
    ref = golden("schedules", f"sched.{sched}.{T}")
    buf = O.schedule_buffers(sched, T)
    for i, n in enumerate(names):
        np.testing.assert_allclose(buf[n], ref[i], rtol=3e-6, atol=1e-30, err_msg=f"{sched}.{T}.{n}")


def test_schedule_known_answers():
    # SURVEY 8a1 probes of the reference
    b = O.schedule_buffers("sigmoid2", 1000)
    assert abs(b["betas"][0] - 6.621860e-07) < 1e-12
    assert abs(b["betas"][499] - 7.816005e-04) < 1e-9
    assert b["betas"][999] == np.float32(0.999)
    assert abs(b["alphas_cumprod"][499] - 0.94498267) < 1e-6
    assert abs(b["posterior_log_variance_clipped"][0] - (-46.0517)) < 1e-3


def test_loss_weight_other_objectives(golden):
    for obj in ("pred_noise", "pred_x0"):
        ref = golden("schedules", f"sched.sigmoid2.1000.loss_weight.{obj}")
        np.testing.assert_allclose(O.schedule_buffers("sigmoid2", 1000, obj)["loss_weight"], ref, rtol=3e-6)


@pytest.mark.parametrize("T,S", [(1000, 50), (1000, 250), (20, 5), (1000, 999)])
def test_ddim_time_grid(golden, T, S):
    ref = golden("schedules", f"ddim_times.{T}.{S}").tolist()
    pairs = O.ddim_time_pairs(T, S)
    assert [p[0] for p in pairs] + [pairs[-1][1]] == ref
    if (T, S) == (1000, 50):
        assert ref[:3] == [999, 979, 959] and ref[-2:] == [19, -1]


def _mod_inputs():
    dim, B, H = 16, 2, 16
    sd = state_dict(dim)
    d = dict(
        sd=sd, dim=dim,
        x=synth.uniform(7, "mod.x", (B, dim, H, H), -1.5, 1.5),
        x2=synth.uniform(7, "mod.x2", (B, 2 * dim, H, H), -1.5, 1.5),
        x8=synth.uniform(7, "mod.x8", (B, 8, H, H), -1.0, 1.0),
        x4=synth.uniform(7, "mod.x4", (B, 4, H, H), -1.5, 1.5),
        xu=synth.uniform(7, "mod.xu", (B, 2 * dim, H // 2, H // 2), -1.5, 1.5),
        temb=synth.uniform(7, "mod.temb", (B, 4 * dim), -1.0, 1.0),
        pos=synth.make_position(B, H, seed=7),
        iso_idx=synth.randint(7, "mod.iso", (B,), 0, 75),
        scale=synth.uniform(7, "mod.scale", (B, dim, 1, 1), -0.5, 0.5),
        shift=synth.uniform(7, "mod.shift", (B, dim, 1, 1), -0.5, 0.5),
    )
    d["iso"] = torch.nn.functional.embedding(d["iso_idx"], sd["iso_embed.weight"]).unsqueeze(1)
    d["pos_emb"] = O.mlp(sd, "pos_mlp", O.learned_sinusoidal_pos_emb(sd, "pos_enc", d["pos"]))
    return d


MODULE_CASES = {
    "pos_enc": lambda m: O.learned_sinusoidal_pos_emb(m["sd"], "pos_enc", m["pos"]),
    "pos_emb": lambda m: m["pos_emb"],
    "time_mlp": lambda m: O.time_mlp(m["sd"], torch.tensor([0, 999]), m["dim"]),
    "time_mlp_mid": lambda m: O.time_mlp(m["sd"], torch.tensor([1, 500]), m["dim"]),
    "block": lambda m: O.block(m["sd"], "downs.0.0.block1", m["x"], 8),
    "block_ss": lambda m: O.block(m["sd"], "downs.0.0.block1", m["x"], 8, (m["scale"], m["shift"])),
    "resnet_same": lambda m: O.resnet_block(m["sd"], "downs.0.0", m["x"], m["temb"], 8),
    "resnet_resconv": lambda m: O.resnet_block(m["sd"], "final_res_block", m["x2"], m["temb"], 8),
    "resnet_g2": lambda m: O.resnet_block(m["sd"], "shot_time", m["x"], m["temb"], 2),
    "resnet_pos": lambda m: O.resnet_block_pos(m["sd"], "pos_block1", m["x"], m["pos_emb"], 2),
    "attn_block": lambda m: O.attn_block(m["sd"], "downs.0.2", m["x"], m["iso"]),
    "mlp_shot1": lambda m: O.mlp(m["sd"], "shot_mlp1", m["x8"]),
    "mlp_shot3": lambda m: O.mlp(m["sd"], "shot_mlp3", m["x"]),
    "downsample": lambda m: O.pixel_unshuffle_conv(m["sd"], "downs.0.3", m["x"]),
    "upsample": lambda m: O.upsample_conv(m["sd"], "ups.2.3", m["xu"]),
    "init_conv": lambda m: O.conv(m["sd"], "init_conv", m["x4"], padding=3),
    "final_conv": lambda m: O.conv(m["sd"], "final_conv", m["x"]),
}


@pytest.fixture(scope="module")
def mod_inputs():
    return _mod_inputs()


@pytest.mark.parametrize("name", sorted(MODULE_CASES))
def test_module_golden(golden, mod_inputs, name):
    with torch.no_grad():
        got = MODULE_CASES[name](mod_inputs).numpy()
    ref = golden("modules", f"mod.{name}")
    assert got.shape == ref.shape
    assert rel_err(got, ref) < TOL, name


def test_attention_extension_goldens(golden):
    from noisediff_amd.spec import attention_param_spec
    C = 128
    xa = synth.uniform(7, "mod.xa", (2, C, 8, 8), -1.5, 1.5)
    sd = synth.make_state_dict(attention_param_spec("mid_attn", C), 0)
    with torch.no_grad():
        got = O.attention(sd, "mid_attn", xa)
        assert rel_err(got.numpy(), golden("modules", "mod.attention")) < TOL
        assert rel_err(got.numpy(), golden("modules", "mod.attention_flash")) < TOL   # SDPA == einsum path
        sdl = {"l.norm.g": torch.ones(1, C, 1, 1), "l.to_qkv.weight": sd["mid_attn.to_qkv.weight"],
               "l.to_out.0.weight": sd["mid_attn.to_out.weight"], "l.to_out.0.bias": sd["mid_attn.to_out.bias"],
               "l.to_out.1.g": synth.uniform(7, "mod.lat_g", (1, C, 1, 1), 0.5, 1.5)}
        assert rel_err(O.linear_attention(sdl, "l", xa).numpy(), golden("modules", "mod.linear_attention")) < TOL
        assert rel_err(O.rms_norm(torch.ones(1, C, 1, 1), xa).numpy(), golden("modules", "mod.rmsnorm")) < TOL


def test_cross_attention_is_a_per_sample_bias(mod_inputs):
    """with a 1-token context, CrossAttention == to_out(to_v(ctx)) for every token."""
    m = mod_inputs
    sd, p = m["sd"], "downs.0.2.attn"
    t = m["x"].permute(0, 2, 3, 1).reshape(2, 256, 16)
    with torch.no_grad():
        full = O.cross_attention(sd, p, t, m["iso"])
        short = O.linear(sd, p + ".to_out.0", O.linear(sd, p + ".to_v", m["iso"]))
    assert float((full - short).abs().max()) < 2e-6


@pytest.mark.parametrize("dim,H", [(16, 32), (32, 64)])
def test_net_forward_golden(golden, dim, H):
    B = 2
    sd = state_dict(dim)
    cond = synth.make_condition(B, H, seed=1)
    x = synth.make_noise(2, "net.x", B, 4, H)
    with torch.no_grad():
        for t in (0, 500, 999):
            taps = {}
            y = O.noisediff_forward(sd, x, torch.full((B,), t, dtype=torch.long), cond, taps=taps)
            assert rel_err(y.numpy(), golden("net", f"net.d{dim}.h{H}.t{t}")) < TOL
            if t == 500:
                for k, v in taps.items():
                    assert rel_err(sub(v), golden("net", f"net.d{dim}.h{H}.t{t}.tap.{k}")) < TOL, k
        y = O.noisediff_forward(sd, x, torch.tensor([3, 777]), cond)
        assert rel_err(y.numpy(), golden("net", f"net.d{dim}.h{H}.tmixed")) < TOL


def _run(dim, B, H, T, S, eta=0.0, return_all=False, sched="sigmoid2", objective="pred_v", mid=False, grab=()):
    sd = state_dict(dim, mid_attn=mid)
    cond = synth.make_condition(B, H, seed=1)
    vs = {}
    n = {"i": 0}

    def on_step(t, img, out):
        if n["i"] in grab:
            vs[n["i"]] = out.numpy().copy()
        n["i"] += 1

    res = O.sample(sd, cond, image_size=H, batch_size=B, timesteps=T, sampling_timesteps=S,
                   beta_schedule_name=sched, objective=objective, eta=eta,
                   x_T=synth.make_noise(2, "x_T", B, 4, H), noise=noise_fn(2, B, 4, H),
                   return_all=return_all, mid_attention="mid_attn" if mid else None, on_step=on_step)
    return res.numpy(), vs


def test_sampler_config1_ddim50(golden):
    """BASELINE config 1: d=32, 64x64x4, 50-step DDIM, batch 4 -- end to end plus pre-clamp v."""
    res, vs = _run(32, 4, 64, 1000, 50, grab={0, 25, 49})
    for k in (0, 25, 49):
        assert rel_err(vs[k], golden("sampler", f"samp.cfg1.v{k}")) < 1e-4, k
    assert rel_err(res, golden("sampler", "samp.cfg1.out")) < 1e-4


def test_sampler_ddpm20(golden):
    res, vs = _run(16, 2, 32, 20, None, grab={0, 10, 19})
    for k in (0, 10, 19):
        assert rel_err(vs[k], golden("sampler", f"samp.ddpm20.v{k}")) < 1e-4
    assert rel_err(res, golden("sampler", "samp.ddpm20.out")) < 1e-4
    # the reference's own x_T injection point (preset_mean) gives the same trajectory
    assert rel_err(res, golden("sampler", "samp.ddpm20_preset.out")) < 1e-4


def test_sampler_return_all_and_eta(golden):
    res, _ = _run(16, 2, 32, 4, None, return_all=True)
    ref = golden("sampler", "samp.ddpm4_all.out")
    assert res.shape == ref.shape == (2, 5, 4, 32, 32)
    assert rel_err(res, ref) < 1e-4
    res, _ = _run(16, 2, 32, 20, 5, eta=0.5, return_all=True)
    ref = golden("sampler", "samp.ddim5_eta.out")
    assert res.shape == ref.shape == (2, 6, 4, 32, 32)
    assert rel_err(res, ref) < 1e-4


def test_sampler_other_objectives(golden):
    res, _ = _run(16, 2, 32, 50, None, sched="linear", objective="pred_noise")
    assert rel_err(res, golden("sampler", "samp.ddpm50_eps_linear.out")) < 1e-4
    res, _ = _run(16, 2, 32, 20, 5, sched="cosine", objective="pred_x0")
    assert rel_err(res, golden("sampler", "samp.ddim5_x0_cosine.out")) < 1e-4


def test_sampler_config4_toy_mid_attention(golden):
    res, vs = _run(16, 2, 64, 1000, 10, mid=True, grab={0, 9})
    for k in (0, 9):
        assert rel_err(vs[k], golden("sampler", f"samp.cfg4toy.v{k}")) < 1e-4
    assert rel_err(res, golden("sampler", "samp.cfg4toy.out")) < 1e-4


def test_noise_call_order_recorded(meta):
    # DDIM: x_T then one randn_like per pair except the last; DDPM: one per step with t > 0
    assert meta["samp.cfg1.calls"] == ["x_T"] + [f"noise.{i}" for i in range(49)]
    assert meta["samp.ddpm20.calls"] == ["x_T"] + [f"noise.{i}" for i in range(19)]


def test_philox_known_answers():
    # Random123 KAT vectors for philox4x32-10
    z = O.philox4x32_10(np.zeros((1, 4), np.uint32), np.zeros((1, 2), np.uint32))[0]
    assert [hex(int(v)) for v in z] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    f = np.full((1, 4), 0xFFFFFFFF, np.uint32)
    z = O.philox4x32_10(f, f[:, :2])[0]
    assert [hex(int(v)) for v in z] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    c = np.array([[0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344]], np.uint32)
    k = np.array([[0xA4093822, 0x299F31D0]], np.uint32)
    z = O.philox4x32_10(c, k)[0]
    assert [hex(int(v)) for v in z] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_lsid_oracle_matches_reference_golden(golden):
    """Next row 8f-1: LSID forward (incl. ceil-mode pooling + crop at an odd size) and the synth->denoise PSNR."""
    from noisediff_amd.spec import lsid_param_spec
    sd = synth.make_state_dict(lsid_param_spec(), 0)
    with torch.no_grad():
        for (B, H, W) in ((2, 64, 64), (1, 36, 44)):
            x = synth.uniform(9, f"lsid.x.{H}x{W}", (B, 4, H, W), 0.0, 1.0)
            assert rel_err(O.lsid_forward(sd, x).numpy(), golden("lsid", f"lsid.{H}x{W}")) < TOL
        clean = synth.uniform(9, "lsid.clean", (2, 4, 64, 64), 0.0, 1.0)
        noise = synth.make_noise(9, "lsid.noise", 2, 4, 64) * 0.1
        _, out, psnr = O.compose_and_denoise(sd, noise, clean)
    assert rel_err(out.numpy(), golden("lsid", "lsid.compose.out")) < TOL
    assert abs(psnr - float(golden("lsid", "lsid.compose.psnr"))) < 1e-3
