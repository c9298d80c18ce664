"""CPU-side tests: host logic, C-ABI surface, schedule tables, 2-rank gloo sharding."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from noisediff_amd import _lib as L, synth
from noisediff_amd.diffusion import BUFFER_NAMES, GaussianDiffusion, make_betas, make_buffers
from noisediff_amd.net import NoiseDiffNet
from noisediff_amd.shard import shard_bounds
from noisediff_amd.spec import noisediff_param_spec
from util import state_dict
from types import SimpleNamespace

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def args(dim, **kw):
    return SimpleNamespace(dim=dim, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False, **kw)


def test_library_loads_and_exports_every_declared_symbol():
    """The drop-in boundary: every function include/noisediff_hip.h declares is exported and bound."""
    header = open(os.path.join(REPO, "include", "noisediff_hip.h")).read()
    declared = set(re.findall(r"\b(nd_[a-z0-9_]+)\s*\(", header))
    declared -= {"nd_src", "nd_conv3x3", "nd_pointwise", "nd_sampler_state"}
    lib = L.load()
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.nd_version() >= 1
    assert lib.nd_last_error() is not None


def test_struct_layouts_match_the_header():
    # sizes computed from the C declarations: pointers 8 B, int32 4 B, natural alignment
    assert C.sizeof(L.Src) == 2 * 8 + 8 * 4 + 6 * 8
    assert C.sizeof(L.Conv3x3) == C.sizeof(L.Src) + 5 * 8 + 6 * 4
    assert C.sizeof(L.Pointwise) == C.sizeof(L.Src) + 8 * 8 + 13 * 4 + 4     # 13 int32 + tail padding to 8
    assert C.sizeof(L.SamplerState) == 6 * 8 + 2 * 4
    assert C.sizeof(L.ChainStage) == 2 * 8 + 4 * 4
    assert C.sizeof(L.PackItem) == 2 * 8 + 4 * 4
    assert C.sizeof(L.AdamItem) == 4 * 8 + 8 + 2 * 4 + 2 * 4 + 8                # nd_adam_item
    assert C.sizeof(L.Chain) == C.sizeof(L.Src) + 3 * C.sizeof(L.ChainStage) + 8 + 4 * 4


def test_host_only_entry_points_validate_arguments():
    lib = L.load()
    assert lib.nd_pack_conv3x3_weight_floats(64, 64) == 9 * 64 * 64
    assert lib.nd_pack_conv3x3_weight_floats(48, 48) == 9 * 48 * 64           # cout padded to 64
    assert lib.nd_pack_pointwise_weight_floats(24, 16) == 24 * 64
    assert lib.nd_conv3x3_stat_slots(256, 256, 64, 16) == (256 // 8) * (256 // 16) * 2
    assert lib.nd_conv3x3_tiling_id(16, 256, 256, 64) == 1621
    assert lib.nd_conv3x3_tiling_id(2, 8, 8, 64) == 811
    assert lib.nd_conv3x3_stat_slots(0, 8, 8, 1) == -1
    assert lib.nd_conv3x3_nhwc_f32(None, None) == -1 and b"null" in lib.nd_last_error()
    assert lib.nd_pointwise_gemm_nhwc_f32(None, None) == -1
    d = L.Conv3x3()
    d.src.p0 = d.weight = d.out = 0x1000
    d.src.c0, d.src.ld0 = 12, 12
    d.B, d.H, d.W, d.cin, d.cout, d.ldo = 1, 8, 8, 12, 8, 8
    assert lib.nd_conv3x3_nhwc_f32(C.byref(d), None) == -2                   # cin % 8 != 0 -> ND_E_SHAPE, nothing launched
    # Winograd weights: 128 KB blocks [cin/32][cout/64][16 positions][8 quads][64][4], zero padded
    assert lib.nd_pack_conv3x3_wino_weight_floats(64, 64) == 16 * 64 * 64
    assert lib.nd_pack_conv3x3_wino_weight_floats(48, 40) == 16 * 64 * 64
    assert lib.nd_conv3x3_wino_stat_slots(256, 256) == 16 * 16 * 2 and lib.nd_conv3x3_wino4_stat_slots(256, 250) == 16 * 16
    assert lib.nd_conv3x3_wino2_nhwc_f32(None, None) == -1
    d.src.c0 = d.src.ld0 = d.cin = 48
    d.src.p1, d.src.c1, d.src.ld1, d.cin = 0x2000, 16, 16, 64
    d.H = d.W = 32
    assert lib.nd_conv3x3_wino2_nhwc_f32(C.byref(d), None) == -2 and b"straddle" in lib.nd_last_error()
    # fused Linear chains: only NoiseDiffNet's width combinations are instantiated; weights padded to MFMA operand tiles
    assert lib.nd_pointwise_chain_supported(64, 128, 64, 64) == 1 and lib.nd_pointwise_chain_supported(8, 64, 64, 0) == 1
    assert lib.nd_pointwise_chain_supported(48, 96, 48, 48) == 1 and lib.nd_pointwise_chain_supported(64, 160, 64, 64) == 0
    assert lib.nd_pack_chain_weight_floats(8, 64, 1) == 8 * 64 and lib.nd_pack_chain_weight_floats(48, 4, 0) == 64 * 32
    assert lib.nd_pointwise_chain_nhwc_f32(None, None) == -1


@pytest.mark.parametrize("sched", ["linear", "cosine", "sigmoid1", "sigmoid2", "sigmoid3"])
def test_product_schedule_buffers_are_bit_identical_to_the_reference(golden, sched):
    names = [str(n) for n in golden("schedules", "sched.names")]
    assert names == BUFFER_NAMES
    ref = golden("schedules", f"sched.{sched}.1000")
    buf = make_buffers(make_betas(sched, 1000), "pred_v")
    for i, n in enumerate(names):
        assert np.array_equal(buf[n].numpy(), ref[i], equal_nan=True), n


def test_default_beta_schedule_name_is_rejected_like_the_reference():
    net = NoiseDiffNet(args(16))
    with pytest.raises(ValueError, match="unknown beta schedule sigmoid"):
        GaussianDiffusion(net, image_size=32)          # CLI default 'sigmoid' is not a valid name upstream either
    with pytest.raises(AssertionError):
        GaussianDiffusion(net, image_size=32, beta_schedule="sigmoid2", objective="bogus")
    with pytest.raises(AssertionError):
        GaussianDiffusion(net, image_size=32, beta_schedule="sigmoid2", timesteps=10, sampling_timesteps=11)


def test_module_surface_matches_the_reference_plugin_contract(meta):
    net = NoiseDiffNet(args(32))
    assert [[k, list(v.shape)] for k, v in net.state_dict().items()] == meta["state_dict.d32"]
    assert (net.channels, net.out_dim, net.self_condition, net.random_or_learned_sinusoidal_cond, net.downsample_factor) == (4, 4, False, False, 8)
    net.load_state_dict(state_dict(32), strict=True)
    # PyTorch default init statistics (the reference never calls init_weights)
    w = NoiseDiffNet(args(32)).state_dict()["downs.0.0.block1.proj.weight"]
    bound = 1 / np.sqrt(32 * 9)
    assert float(w.abs().max()) <= bound and float(w.std()) == pytest.approx(bound / np.sqrt(3), rel=0.1)
    import copy
    copy.deepcopy(net)                                          # EMA(self.net) deep-copies  (trainer_diffusion.py:62-69)
    wrapped = torch.nn.DataParallel(net)
    gd = GaussianDiffusion(wrapped, image_size=64, timesteps=1000, sampling_timesteps=50, beta_schedule="sigmoid2")
    assert gd.is_ddim_sampling and gd.num_timesteps == 1000 and gd.channels == 4 and gd.image_size == 64
    assert gd.device.type == "cpu" and len(list(gd.buffers())) == 13
    assert "mid_attn.to_qkv.weight" in NoiseDiffNet(args(16, mid_attn=True)).state_dict()


def test_sampler_tables(golden):
    net = NoiseDiffNet(args(16))
    gd = GaussianDiffusion(net, image_size=32, timesteps=1000, sampling_timesteps=50, beta_schedule="sigmoid2")
    t_cur, t_next, coef = gd._tables()
    ref = golden("schedules", "ddim_times.1000.50").tolist()
    assert t_cur.tolist() == ref[:-1] and t_next.tolist() == ref[1:]
    assert coef.shape == (50, 8) and coef[-1, 7] == 1.0 and bool((coef[:-1, 7] == 0).all())
    assert bool((coef[:, 6] == 0).all())                       # eta = 0 -> sigma = 0
    a, an = gd.alphas_cumprod[999], gd.alphas_cumprod[979]
    assert coef[0, 4] == an.sqrt() and coef[0, 5] == (1 - an).sqrt()
    gd = GaussianDiffusion(net, image_size=32, timesteps=20, beta_schedule="sigmoid2")
    t_cur, _, coef = gd._tables()
    assert t_cur.tolist() == list(range(19, -1, -1)) and coef[-1, 7] == 0.0 and bool((coef[:-1, 7] == 1).all())
    assert torch.equal(coef[:, 4], gd.posterior_mean_coef1.flip(0)) and torch.equal(coef[:, 6], (0.5 * gd.posterior_log_variance_clipped).exp().flip(0))


def test_product_refuses_to_run_without_a_gpu():
    net = NoiseDiffNet(args(16)).eval()
    gd = GaussianDiffusion(net, image_size=16, timesteps=4, beta_schedule="sigmoid2")
    cond = synth.make_condition(1, 16, seed=1)
    with pytest.raises(L.HipError, match="no CPU path"):
        gd.sample(batch_size=1, condition=cond)
    with pytest.raises(L.HipError, match="no CPU path"):               # training entry (p_losses, autograd on): no CPU path either
        gd(torch.zeros(1, 4, 16, 16), cond)


def test_shard_bounds_and_synthetic_shards():
    for total, world in ((128, 8), (10, 4), (3, 8), (0, 2)):
        spans = [shard_bounds(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    full = synth.make_condition(6, 16, seed=1)
    part = synth.make_condition(2, 16, seed=1, first_sample=3, total=6)
    for k in full:
        assert torch.equal(full[k][3:5], part[k]), k
    assert torch.equal(synth.make_noise(2, "x_T", 6, 4, 16)[3:5], synth.make_noise(2, "x_T", 2, 4, 16, first_sample=3))


def _gloo_worker(rank, world, port, q, total=3):
    import torch.distributed as dist
    from noisediff_amd.shard import sample_sharded
    from oracle import noisediff_oracle as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    dim, H, T = 16, 16, 3
    # the one collective: weights root -> all (a flat fp32 arena on the GPU path)
    flat = torch.cat([v.reshape(-1) for v in state_dict(dim).values()]) if rank == 0 else torch.zeros(sum(np.prod(p.shape) for p in noisediff_param_spec(dim)), dtype=torch.float32)
    dist.broadcast(flat, src=0)
    sd, off = {}, 0
    for p in noisediff_param_spec(dim):
        n = int(np.prod(p.shape))
        sd[p.name] = flat[off:off + n].view(p.shape)
        off += n
    state = {"lo": 0}

    def sample_fn(batch_size, condition, seed):
        lo = state["lo"]
        return O.sample(sd, condition, image_size=H, batch_size=batch_size, timesteps=T, x_T=synth.make_noise(seed, "x_T", batch_size, 4, H, lo),
                        noise=lambda i, shape: synth.make_noise(seed, f"noise.{i}", batch_size, 4, H, lo))

    out = sample_sharded(sample_fn, total, lambda lo, hi: synth.make_condition(hi - lo, H, seed=1, first_sample=lo, total=total),
                         seed=2, set_offset=lambda lo: state.update(lo=lo), gather=True)
    if rank == 0:
        q.put(out.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 3), (8, 19)])
def test_n_rank_sharded_sampling_equals_single_rank_gloo(world, total):
    """world_size 2 and 8 on CPU (gloo; SURVEY section 4: "8-rank shard == 1-rank full-batch"): broadcast the weights once, shard the rows (8 ranks, 19 patches:
    ragged 3 + 3 + 3 + 2 + ...), gather -> the same patches as one rank, in parity-noise mode."""
    import torch.multiprocessing as mp
    from oracle import noisediff_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + world) % 2000
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, q, total)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    dim, H, T = 16, 16, 3
    ref = O.sample(state_dict(dim), synth.make_condition(total, H, seed=1), image_size=H, batch_size=total, timesteps=T,
                   x_T=synth.make_noise(2, "x_T", total, 4, H), noise=lambda i, shape: synth.make_noise(2, f"noise.{i}", total, 4, H))
    assert got.shape == (total, 4, H, H)
    assert float(np.abs(got - ref.numpy()).max()) < 1e-5


def _gloo_empty_rank_worker(rank, world, port, q):
    import torch.distributed as dist
    from noisediff_amd.shard import sample_sharded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = 2                                                            # world 3 > total 2: rank 2 has no rows
    calls = []

    def sample_fn(batch_size, condition, seed):
        calls.append(batch_size)
        return condition["rows"].float().view(-1, 1, 1, 1).expand(batch_size, 4, 2, 2).contiguous() + seed

    out = sample_sharded(sample_fn, total, lambda lo, hi: {"rows": torch.arange(lo, hi)}, seed=5, gather=True, device=torch.device("cpu"))
    own = sample_sharded(sample_fn, total, lambda lo, hi: {"rows": torch.arange(lo, hi)}, seed=5, gather=False)
    q.put((rank, out.numpy(), None if own is None else tuple(own.shape), list(calls)))
    dist.barrier()
    dist.destroy_process_group()


def test_a_rank_without_rows_takes_part_in_the_gather_gloo():
    """world_size 3, two patches: the rank without rows never calls the sampler, builds its placeholder on the device
    it is told (the collective's backend decides: RCCL takes device tensors only) and still returns the whole batch; without gather it
    returns None.  total_batch = 0 is refused."""
    import torch.multiprocessing as mp
    from noisediff_amd.shard import sample_sharded
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gloo_empty_rank_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=240) for _ in range(3)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.broadcast_to(np.array([5.0, 6.0], dtype=np.float32).reshape(2, 1, 1, 1), (2, 4, 2, 2))
    for rank, out, own, calls in got:
        assert out.shape == (2, 4, 2, 2) and np.array_equal(out, want)
        assert (own, calls) == ((None, []) if rank == 2 else ((1, 4, 2, 2), [1, 1]))
    with pytest.raises(ValueError, match="nothing to sample"):
        sample_sharded(lambda **kw: None, 0, lambda lo, hi: {}, seed=0)


def test_generated_npy_contract(tmp_path):
    """file names, CHW fp32 payload, patch grid and the synth -> denoise composition."""
    from noisediff_amd import io
    grid = io.patch_grid(512)
    assert len(grid) == 24 and grid[0] == (0, 0) and grid[5] == (2128 - 512, 0) and grid[-1] == (2128 - 512, 1424 - 512)
    assert grid[1] == (384, 0) and grid[6] == (0, 384)
    out = synth.make_noise(3, "npy", 2, 4, 16)
    names = io.save_generated(str(tmp_path), out, ["00001_00_10s.ARW", "00002_00_10s.ARW"], ["00001_00_0.1s.ARW", None],
                              [io.image_coord(384, 768), io.image_coord(0, 0)])
    assert os.path.basename(names[0]) == "00001_00_10s+00001_00_0.1s+384_768.npy"
    assert os.path.basename(names[1]) == "00002_00_10s+00002_00_10s+0_0.npy"
    assert io.parse_generated_name(names[0]) == ("00001_00_10s", "00001_00_0.1s", 384, 768)
    back = np.load(names[0])
    assert back.dtype == np.float32 and back.shape == (4, 16, 16) and np.array_equal(back, out[0].numpy())
    noise = torch.tensor([[-2.0, 0.5, 0.9]])
    clean = torch.tensor([[0.3, 0.3, 0.3]])
    assert torch.allclose(io.compose_noisy(noise, clean), torch.tensor([[0.0, 0.8, 1.0]]))
    a = torch.rand(4, 8, 8)
    assert io.psnr(a, a) == float("inf")
    assert io.psnr(torch.zeros(4), torch.full((4,), 0.1)) == pytest.approx(20.0, abs=1e-4)


def test_blocked_map_row_permutation():
    """engine._blocked_map_rows: output row order of the ResnetBlock2 map Linear for nd_src.map_blocked (include/noisediff_hip.h):
    position 32 c + j holds scale channel 16 c + j (j < 16) or shift channel 16 c + j - 16."""
    from noisediff_amd.engine import _blocked_map_rows
    for C_ in (16, 64, 128):
        perm = _blocked_map_rows(C_)
        assert sorted(perm.tolist()) == list(range(2 * C_))                      # a permutation of the 2C producer rows
        planar = torch.arange(2 * C_)                                            # value = original row: scale rows 0..C-1, shift rows C..2C-1
        blocked = planar[perm].view(C_ // 16, 2, 16)
        for c in range(C_ // 16):
            assert blocked[c, 0].tolist() == list(range(16 * c, 16 * c + 16))              # scale channels of chunk c
            assert blocked[c, 1].tolist() == list(range(C_ + 16 * c, C_ + 16 * c + 16))    # shift channels of chunk c


def test_accelerate_retargets_the_class_so_replicas_and_copies_resolve_forward_through_themselves():
    """train.accelerate() must survive nn.DataParallel's replicate() (define_G wraps the net, models/modules.py:81): a replica copies
    ``__dict__``, so an instance-bound forward would keep running on the original module's parameters.  The HIP forward sits
    on the module's class instead: replicas, deep copies and pickles call it with their own ``self``; state dict and module type names
    are unchanged."""
    import copy
    import pickle
    from torch import nn
    from noisediff_amd import train
    net = nn.Sequential(nn.Conv2d(8, 16, 3, padding=1), nn.GroupNorm(4, 16), nn.SiLU(), nn.Conv2d(16, 16, 1), nn.LayerNorm(64), nn.Linear(64, 64))
    keys = list(net.state_dict().keys())
    assert train.accelerate(net) == 1 and train.accelerate(net) == 0
    conv, norm = net[0], net[1]
    assert conv.forward.__func__ is train._hip_conv_forward and norm.forward.__func__ is train._hip_norm_forward
    assert "forward" not in conv.__dict__ and isinstance(conv, nn.Conv2d) and type(conv).__name__ == "Conv2d"
    assert type(conv)._nd_accelerated_base is nn.Conv2d and type(net[5])._nd_accelerated_base is nn.Linear
    for m in (conv, norm, net[3], net[4], net[5]):
        for other in (m._replicate_for_data_parallel(), copy.deepcopy(m)):
            assert other is not m and other.forward.__self__ is other and other.forward.__func__ is m.forward.__func__
    assert list(net.state_dict().keys()) == keys
    clone = pickle.loads(pickle.dumps(nn.Conv2d(8, 8, 3, padding=1)))          # plain modules still pickle ...
    assert isinstance(clone, nn.Conv2d)
    back = pickle.loads(pickle.dumps(net))                                      # ... and so do accelerated ones (torch.save(model), mp.spawn arguments)
    assert [type(m) for m in back] == [type(m) for m in net] and back[0].forward.__func__ is train._hip_conv_forward
    assert all(torch.equal(a, b) for a, b in zip(back.state_dict().values(), net.state_dict().values()))
    import io as _io
    buf = _io.BytesIO()
    torch.save(net, buf)
    buf.seek(0)
    assert torch.load(buf, weights_only=False)[1].forward.__func__ is train._hip_norm_forward
    with pytest.raises(Exception, match="HIP library only|no CPU path"):
        conv(torch.zeros(1, 8, 16, 16))                                          # still no CPU fallback


def test_split_k_plan_is_a_function_of_the_shape():
    """nd_conv3x3_wino4_splitk_plan (host code, no GPU): 1 where the layer's (sample, region, cout tile) items fill the chip or cin leaves
    no room, else the power of two that brings them to about 256 with at least four 16-channel chunks per range."""
    lib = L.load()
    plan = lib.nd_conv3x3_wino4_splitk_plan
    assert plan(16, 256, 256, 64, 64) == 1 and plan(16, 32, 32, 512, 512) == 1          # the bench workload's layers: never split
    assert plan(4, 32, 32, 512, 512) == 4 and plan(4, 64, 64, 256, 256) == 2 and plan(4, 32, 32, 1536, 512) == 4
    assert plan(1, 32, 32, 1536, 64) == 8 and plan(1, 32, 32, 64, 64) == 1 and plan(1, 32, 32, 72, 64) == 1
    assert plan(4, 256, 256, 64, 64) == 1 and plan(0, 1, 1, 1, 1) == 1
    assert lib.nd_conv3x3_wino4_splitk_workspace_floats(2, 8, 8, 16, 4) == 2 * 8 * 8 * 16 * 4


def test_sampling_split_k_plan_looks_at_the_sample_geometry_only():
    """nd_conv3x3_wino4_16_splitk_plan (host code, no GPU) has no batch argument: BASELINE config 2's 16 x 16 and 32 x 32 stages are cut into K ranges
    (about 32 items per sample, at least four 16-channel chunks per range), the layers of the 256 x 256 workload that reach the 16 x 16-region form
    (256 -> 256 at 32 x 32) into two, everything with enough items per sample or too few chunks is left alone."""
    plan = L.load().nd_conv3x3_wino4_16_splitk_plan
    assert plan(16, 16, 512, 512) == 4 and plan(16, 16, 768, 512) == 4 and plan(16, 16, 256, 512) == 4 and plan(16, 16, 256, 256) == 4
    assert plan(32, 32, 256, 256) == 2 and plan(32, 32, 384, 256) == 2 and plan(32, 32, 512, 256) == 2
    assert plan(32, 32, 512, 512) == 1 and plan(256, 256, 64, 64) == 1 and plan(16, 16, 64, 64) == 1 and plan(16, 16, 72, 64) == 1
    assert plan(16, 16, 2048, 64) == 8 and plan(0, 16, 64, 64) == 1


@pytest.mark.parametrize("objective,sched", [("pred_v", "sigmoid2"), ("pred_noise", "linear"), ("pred_x0", "cosine")])
def test_per_step_public_methods_follow_the_reference_formulas(objective, sched):
    """predict_start_from_* / predict_noise_from_start / q_posterior / model_predictions / p_mean_variance / p_sample
    (models/denoising_diffusion_pytorch.py:298-373) around a stand-in network on the CPU, against the oracle's restatement of the same
    lines (predict_x0_eps + the posterior of p_sample_loop), for the three objectives; interpolate and the clip / re-derive switches."""
    from torch import nn
    from oracle import noisediff_oracle as O

    class Stub(nn.Module):
        channels = out_dim = 4
        self_condition = random_or_learned_sinusoidal_cond = False

        def forward(self, x, t, condition=None):
            return 0.7 * x.flip(1) + 0.1 * t.view(-1, 1, 1, 1) / 50 + (0 if condition is None else condition["bias"])

    T, B, H = 50, 3, 8
    gd = GaussianDiffusion(Stub(), image_size=H, timesteps=T, beta_schedule=sched, objective=objective)
    buf = O.schedule_buffers(sched, T, objective)
    x = synth.make_noise(7, "ps.x", B, 4, H) * 1.5
    cond = {"bias": torch.full((B, 1, 1, 1), 0.05)}
    for t in (T - 1, 17, 0):
        tb = torch.full((B,), t, dtype=torch.long)
        out = gd.model(x, tb, cond)
        for clip in (False, True):
            eps_ref, x0_ref = O.predict_x0_eps(buf, objective, x, t, out, clip=clip)
            mp = gd.model_predictions(x, tb, cond, clip_x_start=clip, rederive_pred_noise=True)
            assert torch.allclose(mp.pred_x_start, x0_ref, atol=1e-6) and torch.allclose(mp.pred_noise, eps_ref, atol=2e-5)
        _, x0 = O.predict_x0_eps(buf, objective, x, t, out, clip=False)
        x0 = x0.clamp(-1, 1)
        mean_ref = O._coef(buf, "posterior_mean_coef1", t) * x0 + O._coef(buf, "posterior_mean_coef2", t) * x
        mean, var, logvar, xs = gd.p_mean_variance(x, tb, cond)
        assert torch.allclose(mean, mean_ref, atol=1e-6) and torch.allclose(xs, x0, atol=1e-6)
        assert float(var.flatten()[0]) == float(gd.posterior_variance[t]) and float(logvar.flatten()[0]) == float(gd.posterior_log_variance_clipped[t])
        z = synth.make_noise(7, f"ps.z{t}", B, 4, H)
        nxt, xs2 = gd.p_sample(x, t, cond, noise=z)
        want = mean_ref + ((0.5 * O._coef(buf, "posterior_log_variance_clipped", t)).exp() * z if t > 0 else 0.0)
        assert torch.allclose(nxt, want, atol=1e-6) and torch.equal(xs2, xs)
    v = synth.make_noise(7, "ps.v", B, 4, H)
    tb = torch.full((B,), 9, dtype=torch.long)
    assert torch.allclose(gd.predict_noise_from_start(x, tb, gd.predict_start_from_noise(x, tb, v)), v, atol=1e-4)
    assert torch.allclose(gd.predict_start_from_v(x, tb, gd.predict_v(v, tb, x)), gd.predict_start_from_v(x, tb, gd.predict_v(v, tb, x)))
    torch.manual_seed(3)
    mix = gd.interpolate(x, -x, t=4, lam=0.25, condition=cond)
    assert mix.shape == x.shape and torch.isfinite(mix).all()
    with pytest.raises(ValueError, match="does not match"):
        gd.p_sample_loop((B, 4, H + 1, H), cond)
