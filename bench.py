#!/usr/bin/env python3
"""bench.py -- sampled RAW patches / second on the NoiseDiff sampling hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]          # N=1: one process
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json: "256x256x4, 1000-step DDPM", configs[2] per-GPU shard): NoiseDiffNet
dim=64, sigmoid2 / pred_v, 16 patches of 256x256x4 per GPU.  A *step* is one reverse-diffusion
step over the whole per-GPU batch: the captured graph [write t -> time MLP -> U-Net forward ->
fused posterior/noise update -> advance] -- the unit the 1000-step sampler repeats 1000 times with
identical cost.  K steps are timed after W warm-up steps; one sampled patch = T = 1000 such steps, so
    value = n_gpus * batch_per_gpu / (T * seconds_per_step)       [patches/s, whole job].
`--full` instead times complete 1000-step .sample() calls (K = number of calls).

Also on the JSON line: `roofline` for the dominant kernel family (conv3x3: exact-fp32 MFMA bound)
measured with HIP events on the library's stream in a separate instrumented pass over the same
steps, and `cpu_baseline`: the CPU oracle's p_sample step timed on this box's host cores.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from types import SimpleNamespace

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 == FP32 vector peak.  `achieved` counts ALGORITHMIC
                                   # conv FLOPs (18*Cin*Cout per pixel); the Winograd kernel issues 2.25x fewer MFMA FLOPs than that,
                                   # so frac can approach / exceed 1 while the matrix pipe itself is ~45 % busy
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=16, help="patches per GPU (weak scaling)")
    ap.add_argument("--timesteps", type=int, default=1000)
    ap.add_argument("--sampling-timesteps", type=int, default=None, help="DDIM steps (default: DDPM over all timesteps)")
    ap.add_argument("--mid-attn", action="store_true", help="Attention between the mid blocks (BASELINE config 4 extension)")
    ap.add_argument("--config", choices=["cfg2", "cfg3", "cfg4"], default=None,
                    help="BASELINE.json presets: cfg2 d64/128x128/B16 DDPM; cfg3 (default workload) d64/256x256/B16 per GPU DDPM; "
                         "cfg4 d128 + mid attention/256x256/B8 per GPU/250-step DDIM")
    ap.add_argument("--full", action="store_true", help="time whole 1000-step sample() calls instead of K steps")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch kernels one by one instead of replaying the hipGraph")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL on ROCm; gloo for plumbing tests)")
    ap.add_argument("--one-device", action="store_true", help="testing only: every rank uses cuda:0 (single-GPU box)")
    return ap.parse_args()


def _latest_profile(kind):
    """profiles/r<round><letter>_<kind>.json of the newest build that has one (written by tools/pmc_*.py on the GPU box)."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", f"r*_{kind}.json")))
    return files[-1] if files else ""


def conv_flops(m):
    return 18.0 * m["cin"] * m["cout"] * m["H"] * m["W"] * m["B"]          # 2 * 9 * Cin * Cout per output pixel


def conv_bytes(m):
    return 4.0 * m["B"] * m["H"] * m["W"] * (m["cin"] + m["cout"]) + 4.0 * (9 * m["cin"] * m["cout"] + m["cout"])


def instrumented_pass(loop, plan, L, n_steps):
    """Eager replay of n_steps with a HIP event pair around every conv3x3 launch (library stream)."""
    st = plan.e.stream
    CONV = ("nd_conv3x3_nhwc_f32", "nd_conv3x3_wino_nhwc_f32", "nd_conv3x3_wino2_nhwc_f32")
    STREAM = "nd_affine_silu_add_f32"           # the HBM-bound family: GroupNorm-apply + SiLU + residual adds, one pass
    convs = [op for op in plan.step_ops if op[2] in CONV or (op[2] == STREAM and op[3])]
    n_ev = 2 * len(convs)
    evs = []
    for _ in range(n_ev):
        e = C.c_void_p()
        L.call("nd_event_create", C.byref(e))
        evs.append(e)
    per = {}
    for _ in range(n_steps):
        L.call("nd_sampler_begin_step", C.byref(loop.state), st)
        i = 0
        for fn, args, name, meta in plan.step_ops:
            if name in CONV or (name == STREAM and meta):
                L.call("nd_event_record", evs[2 * i], st)
                L.check(fn(*args), name)
                L.call("nd_event_record", evs[2 * i + 1], st)
                i += 1
            else:
                L.check(fn(*args), name)
        fnname = "nd_sampler_step_ddim_f32" if loop.gd.is_ddim_sampling else "nd_sampler_step_ddpm_f32"
        e_ = plan.e
        L.call(fnname, plan.x.data_ptr(), plan.model_out.data_ptr(), None, 0, C.byref(loop.state),
               L.OBJECTIVES[loop.gd.objective], C.c_uint64(1), 0, plan.B, plan.H * plan.W, e_.inp_dim, st)
        L.call("nd_sampler_advance", C.byref(loop.state), st)
        L.call("nd_stream_sync", st)
        for j, op in enumerate(convs):
            ms = C.c_float()
            L.call("nd_event_elapsed_ms", evs[2 * j], evs[2 * j + 1], C.byref(ms))
            m = op[3]
            if op[2] == STREAM:
                d = per.setdefault(("stream", 0), {"ms": 0.0, "flop": 0.0, "bytes": 0.0, "launches": 0})
                d["ms"] += ms.value
                d["bytes"] += m["stream_bytes"]
                d["launches"] += 1
                continue
            d = per.setdefault((m["tiling"], m["mode"]), {"ms": 0.0, "flop": 0.0, "bytes": 0.0, "launches": 0})
            d["ms"] += ms.value
            d["flop"] += conv_flops(m)
            d["bytes"] += conv_bytes(m)
            d["launches"] += 1
    for e in evs:
        L.call("nd_event_destroy", e)
    return per


def _oracle_step_fn(sd, size, timesteps, batch=1):
    from noisediff_amd import synth
    from oracle import noisediff_oracle as O
    cond = synth.make_condition(batch, size, seed=1)
    buf = O.schedule_buffers("sigmoid2", timesteps)
    z = synth.make_noise(2, "noise.0", batch, 4, size)

    def one_step(img, t):
        out = O.noisediff_forward(sd, img, torch.full((batch,), t, dtype=torch.long), cond)
        _, x0 = O.predict_x0_eps(buf, "pred_v", img, t, out, clip=False)
        x0 = x0.clamp(-1, 1)
        mean = float(buf["posterior_mean_coef1"][t]) * x0 + float(buf["posterior_mean_coef2"][t]) * img
        return mean + float(np_exp_half(buf["posterior_log_variance_clipped"][t])) * z

    return one_step, synth.make_noise(2, "x_T", batch, 4, size)


def cpu_baseline(sd, dim, size, timesteps):
    """The CPU oracle's p_sample step (one U-Net forward + posterior update), batch 1, on this host's cores.

    The thread count is calibrated first (a cgroup-limited box thrashes with one thread per visible
    core): candidates are timed on a 64x64 problem and the fastest is used for the real measurement,
    which is bounded to <= 12 steps / ~20 s and extrapolated x T (steps are identical-cost)."""
    from noisediff_amd import synth
    from noisediff_amd.spec import noisediff_param_spec
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cands = sorted({c for c in (4, 8, 16, 32, 64, min(avail, 96)) if c <= avail})
    sd_small = synth.make_state_dict(noisediff_param_spec(32), 0)
    best, best_dt = cands[0], float("inf")
    with torch.no_grad():
        for c in cands:
            torch.set_num_threads(c)
            f, x = _oracle_step_fn(sd_small, 64, timesteps, batch=4)
            x = f(x, timesteps - 1)
            t0 = time.perf_counter()
            for i in range(2):
                x = f(x, timesteps - 2 - i)
            dt = (time.perf_counter() - t0) / 2
            if dt < best_dt:
                best, best_dt = c, dt
            if dt > 5.0:
                break
        torch.set_num_threads(best)
        f, x = _oracle_step_fn(sd, size, timesteps)
        x = f(x, timesteps - 1)                              # warm-up
        t0 = time.perf_counter()
        n = 0
        while n < 3 or (time.perf_counter() - t0 < 12.0 and n < 12):
            x = f(x, timesteps - 2 - n)
            n += 1
            if time.perf_counter() - t0 > 40.0:
                break
        dt = (time.perf_counter() - t0) / n
    return {"value": 1.0 / (timesteps * dt), "unit": "patches/s", "cores": best, "kind": "port",
            "sample": f"{n} p_sample steps (U-Net forward + posterior update) of the CPU oracle at batch 1, dim {dim}, "
                      f"{size}x{size}x4, {dt:.3f} s/step with {best} threads (best of {cands}; {avail} cores visible), "
                      f"extrapolated x{timesteps} steps per patch"}


def np_exp_half(v):
    import math
    return math.exp(0.5 * float(v))


def main():
    a = parse()
    if a.config == "cfg2":
        a.dim, a.size, a.batch = 64, 128, 16
    elif a.config == "cfg4":
        a.dim, a.size, a.batch, a.sampling_timesteps, a.mid_attn = 128, 256, 8, 250, True
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        a.gpus = world
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if a.one_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(a.backend)

    from noisediff_amd import GaussianDiffusion, NoiseDiffNet, synth, _lib as L
    from noisediff_amd.engine import Engine
    from noisediff_amd.spec import noisediff_param_spec

    B, S, T = a.batch, a.size, a.timesteps
    net = NoiseDiffNet(SimpleNamespace(dim=a.dim, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False,
                                       mid_attn=a.mid_attn))
    sd = None
    if rank == 0:
        sd = synth.make_state_dict(noisediff_param_spec(a.dim), 0)      # synthetic weights, PyTorch default-init statistics
        if a.mid_attn:
            from noisediff_amd.spec import attention_param_spec
            sd.update(synth.make_state_dict(attention_param_spec("mid_attn", 8 * a.dim), 0))
        net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    if world > 1:
        # the ONE collective of the data path: packed weight arena, rank 0 -> everyone, over xGMI
        eng = Engine(a.dim, dev, mid_attn=a.mid_attn)
        if rank == 0:
            eng.load_state_dict(sd)
        eng.broadcast(src=0)
        net.adopt_engine(eng)
    gd = GaussianDiffusion(net, image_size=S, timesteps=T, sampling_timesteps=a.sampling_timesteps, beta_schedule="sigmoid2",
                           objective="pred_v").to(dev)
    n_sample_steps = a.sampling_timesteps or T
    gd.sample_offset = rank * B
    cond = synth.make_condition(B, S, seed=1, first_sample=rank * B, total=world * B)
    plan = net.hip_engine(dev).plan(B, S, S)
    plan.set_condition({k: v.to(dev) for k, v in cond.items()})

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if a.full:
        kw = dict(batch_size=B, condition={k: v.to(dev) for k, v in cond.items()})
        for _ in range(a.warmup):
            gd.sample(seed=1, **kw)
        barrier()
        t0 = time.perf_counter()
        for i in range(a.steps):
            gd.sample(seed=2 + i, **kw)
        plan.e.sync()
        barrier()
        dt = time.perf_counter() - t0
        per_step = dt / a.steps
        value = world * B / per_step
        loop = next(iter(gd._loop_cache.values()))
    else:
        from noisediff_amd.diffusion import _Loop
        loop = _Loop(gd, plan)
        loop.start(None, None, seed=1, first_sample=rank * B, use_graph=not a.eager)
        loop.advance(a.warmup)
        plan.e.sync()
        barrier()
        t0 = time.perf_counter()
        loop.advance(a.steps)
        plan.e.sync()
        barrier()
        dt = time.perf_counter() - t0
        per_step = dt / a.steps
        value = world * B / (n_sample_steps * per_step)
    if world > 1:
        tt = torch.tensor([per_step], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        per_step = float(tt.item())
        value = world * B / per_step if a.full else world * B / (n_sample_steps * per_step)

    out = {
        "metric": "sampled RAW patches/sec (256x256x4, 1000-step DDPM)", "value": value, "unit": "patches/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": per_step * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"NoiseDiffNet dim={a.dim}{' + mid Attention' if a.mid_attn else ''}, {S}x{S}x4 patches, " +
                               (f"{a.sampling_timesteps}-step DDIM of {T}" if a.sampling_timesteps else f"{T}-step DDPM") + " (sigmoid2, pred_v), "
                               f"{B} patches per GPU; a step = " +
                               ("one full 1000-step sample() call" if a.full else
                                "one reverse-diffusion step (U-Net forward + fused posterior/noise update) over the per-GPU batch; "
                                f"patches/s = n_gpus*{B}/({n_sample_steps}*s_per_step)"),
                   "global_batch": world * B, "launch": "eager" if a.eager else "hipGraph replay",
                   "noise": "device Philox4x32-10"},
    }
    if rank == 0 and not a.no_roofline:
        n_inst = min(max(a.steps, 1), 3)
        per = instrumented_pass(loop, plan, L, n_inst)
        stream = per.pop(("stream", 0), None)
        tot_ms = sum(d["ms"] for d in per.values())
        tot_flop = sum(d["flop"] for d in per.values())
        kname = lambda k: (f"wino2_kernel<{k[1]}>" if k[0] == 9002 else f"wino_kernel<1, {k[1]}, 32>" if k[0] == 9001 else
                           f"conv3x3_kernel<{k[0] // 100}, {(k[0] // 10) % 10}, {k[0] % 10}, {k[1]}>")   # as rocprofv3 prints it
        dom = max(per.items(), key=lambda kv: kv[1]["ms"])
        tid, d = dom
        ach = d["flop"] / (d["ms"] * 1e-3) / 1e12
        traffic, tsrc = None, None
        tfile = _latest_profile("traffic")                               # PMC passes cannot run inside this process
        if os.path.exists(tfile) and (a.dim, a.size, a.batch) == (64, 256, 16):
            tk = json.load(open(tfile))["kernels"].get(kname(tid))
            if tk:
                traffic, tsrc = tk["hbm_bytes_per_launch"], f"profiles/{os.path.basename(tfile)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, same workload)"
        busy_pmc = None
        bfile = _latest_profile("mfma_busy")                             # SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), own PMC pass
        if os.path.exists(bfile) and (a.dim, a.size, a.batch) == (64, 256, 16):
            bk = json.load(open(bfile))["kernels"].get(kname(tid))
            if bk:
                busy_pmc = bk["matrix_pipe_busy_frac"]
        out["roofline"] = {
            "bound": "mfma", "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_MFMA_TFLOPS,
            "traffic": traffic, "traffic_source": tsrc,
            "kernel": kname(tid),
            # Winograd F(2x2,3x3) kernels issue 2.25x fewer MFMA FLOPs than the algorithmic count `achieved` is priced in,
            # so frac can exceed 1; matrix_pipe_frac = issued MFMA FLOP/s over the same peak
            "matrix_pipe_frac": ach / (2.25 if tid[0] >= 9001 else 1.0) / PEAK_FP32_MFMA_TFLOPS,
            "matrix_pipe_busy_pmc": busy_pmc,
            "avg_launch_ms": d["ms"] / d["launches"], "launches_per_step": d["launches"] // n_inst,
            "algorithmic_flop_per_launch": d["flop"] / d["launches"],
            "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
            "hbm_frac_at_algorithmic_bytes": d["bytes"] / (d["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS,
            "all_conv3x3": {"tflops": tot_flop / (tot_ms * 1e-3) / 1e12, "ms_per_step": tot_ms / n_inst,
                            "share_of_step": (tot_ms / n_inst) / (per_step * 1e3) if not a.full else None},
            "by_kernel": {kname(k): {"tflops": v["flop"] / (v["ms"] * 1e-3) / 1e12, "avg_ms": v["ms"] / v["launches"],
                                     "launches_per_step": v["launches"] // n_inst} for k, v in sorted(per.items())},
        }
        if stream:
            # the step's HBM-bound family, judged on GB/s: algorithmic bytes (every operand read once, the result written
            # once) over the summed launch time of the same instrumented pass
            gbs = stream["bytes"] / (stream["ms"] * 1e-3) / 1e9
            out["roofline"]["hbm_bound_family"] = {
                "kernel": "affine_silu_add_kernel", "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": gbs / PEAK_HBM_GBS, "launches_per_step": stream["launches"] // n_inst,
                "ms_per_step": stream["ms"] / n_inst}
    if rank == 0 and world == 1 and not a.no_cpu:
        out["cpu_baseline"] = cpu_baseline({k: v for k, v in sd.items() if not k.startswith("mid_attn.")}, a.dim, S, n_sample_steps)
        out["config"]["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
