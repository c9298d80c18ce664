#!/usr/bin/env python3
"""BASELINE config 5 end to end at its stated size: sharded noise synthesis (NoiseDiffNet, DDIM) on 512x512 SID-shaped
synthetic RAW -> noisy = clip(clip(noise,-1,1) + clean, 0, 1) -> LSID denoiser forward -> PSNR, all on the HIP library,
with the same pipeline on the CPU oracle beside it for the first patch of rank 0 (PSNR difference and max error).

    python tools/config5_e2e.py [--size 512] [--dim 64] [--steps 8] [--batch 4]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/config5_e2e.py --batch 4   # N GPUs

One process per GPU; weights are broadcast once (noisediff_amd.shard.broadcast_weights), every rank samples and denoises
its own rows of the global batch (batch per GPU x world), and the per-patch PSNRs are gathered at the end.
Writes gpurun_out/config5_e2e.json on rank 0."""
import argparse, json, os, sys, time
from types import SimpleNamespace
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import torch.distributed as dist
from noisediff_amd import GaussianDiffusion, LSID, NoiseDiffNet, io, shard, synth
from noisediff_amd.spec import lsid_param_spec, noisediff_param_spec

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=512); ap.add_argument("--dim", type=int, default=64)
ap.add_argument("--steps", type=int, default=8, help="DDIM steps of the 1000-step schedule")
ap.add_argument("--batch", type=int, default=4, help="patches per GPU")
ap.add_argument("--backend", default="nccl"); ap.add_argument("--no-oracle", action="store_true")
ap.add_argument("--one-device", action="store_true", help="rehearsal on a one-GPU box: every rank uses cuda:0 (with --backend gloo)")
a = ap.parse_args()
rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
if a.one_device:
    local = 0
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(a.backend, **({"device_id": dev} if a.backend == "nccl" else {}))
S, total = a.size, a.batch * world
args = SimpleNamespace(dim=a.dim, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False)
net = NoiseDiffNet(args)
sd_net = synth.make_state_dict(noisediff_param_spec(a.dim), 0)
sd_lsid = synth.make_state_dict(lsid_param_spec(), 0)
if rank == 0:
    net.load_state_dict(sd_net, strict=True)
net = net.to(dev).eval()
if world > 1:
    shard.broadcast_weights(net, dev, src=0)
den = LSID(SimpleNamespace())
den.load_state_dict(sd_lsid, strict=True)          # 31 MB: every rank loads its own copy
den = den.to(dev).eval()
gd = GaussianDiffusion(net, image_size=S, timesteps=1000, sampling_timesteps=a.steps, beta_schedule="sigmoid2", objective="pred_v").to(dev)

def make_condition(lo, hi):
    c = synth.make_condition(hi - lo, S, seed=1, first_sample=lo, total=total)
    return {k: v.to(dev) for k, v in c.items()}

lo, hi = shard.shard_bounds(total, rank, world)
cond = make_condition(lo, hi)
x_T = synth.make_noise(2, "x_T", total, 4, S)[lo:hi]
steps = torch.stack([synth.make_noise(2, f"noise.{i}", total, 4, S)[lo:hi] for i in range(a.steps - 1)])
torch.cuda.synchronize(dev)
t0 = time.time()
with torch.inference_mode():
    noise = gd.sample(batch_size=hi - lo, condition=cond, noise={"x_T": x_T, "steps": steps})
    noisy = io.compose_noisy(noise, cond["clean_img"])
    out = den(noisy)
torch.cuda.synchronize(dev)
dt = time.time() - t0
psnr = torch.tensor([io.psnr(out[i:i + 1].cpu(), cond["clean_img"][i:i + 1].cpu()) for i in range(hi - lo)], dtype=torch.float64)
if world > 1:
    parts = [torch.zeros_like(psnr) for _ in range(world)]
    dist.all_gather(parts, psnr.to(dev) if a.backend == "nccl" else psnr)
    psnr = torch.cat([p.cpu() for p in parts])
res = {"config": f"NoiseDiffNet d={a.dim}, {a.steps}-step DDIM -> compose -> LSID, {S}x{S}x4, {a.batch} patches per GPU x {world} GPU(s)",
       "psnr_db_per_patch": [round(float(v), 4) for v in psnr], "pipeline_seconds_rank0": dt}
if rank == 0 and not a.no_oracle:
    from oracle import noisediff_oracle as O
    c1 = {k: v[:1].cpu() for k, v in cond.items()}
    t0 = time.time()
    with torch.no_grad():
        ref_noise = O.sample(sd_net, c1, image_size=S, batch_size=1, timesteps=1000, sampling_timesteps=a.steps, x_T=x_T[:1],
                             noise=lambda i, shape: steps[i][:1])
        _, ref_out, ref_psnr = O.compose_and_denoise(sd_lsid, ref_noise, c1["clean_img"])
    res["oracle_first_patch"] = {"psnr_db": ref_psnr, "hip_psnr_db": float(psnr[0]), "psnr_abs_diff_db": abs(ref_psnr - float(psnr[0])),
                                 "noise_max_abs_err": float((noise[:1].cpu() - ref_noise).abs().max()),
                                 "denoised_max_abs_err": float((out[:1].clamp(0, 1).cpu() - ref_out).abs().max()),
                                 "oracle_seconds": time.time() - t0}
if rank == 0:
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(REPO, "gpurun_out", "config5_e2e.json"), "w"), indent=1)
    print(json.dumps(res), flush=True)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
