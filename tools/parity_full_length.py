#!/usr/bin/env python3
"""Full-length parity at the headline configuration: ONE patch, d=64, 256x256x4, 1000-step DDPM (sigmoid2, pred_v), the HIP
sampler against the CPU oracle on identical weights, conditions, x_T and per-step noise (the explicit-noise parity mode of
GaussianDiffusion.sample).  The committed goldens cover 50-step DDIM and 20-step DDPM; this shows what 1000 chained steps do to
the difference.  Takes several minutes of host time (the oracle runs ~0.4 s per step); progress is printed every 50 steps.
usage: python tools/parity_full_length.py [--size 256] [--steps 1000] [--dim 64] -> gpurun_out/parity_full_length.json"""
import argparse, json, os, sys, time
from types import SimpleNamespace
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from noisediff_amd import GaussianDiffusion, NoiseDiffNet, synth
from noisediff_amd.spec import noisediff_param_spec
from oracle import noisediff_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256); ap.add_argument("--steps", type=int, default=1000)
ap.add_argument("--dim", type=int, default=64); ap.add_argument("--threads", type=int, default=16)
ap.add_argument("--oracle-dtype", choices=["f32", "f64"], default="f32",
                help="f64: the oracle's arithmetic in float64 (torch default dtype) -- the yardstick that shares no rounding with either product form; about nine times the host time")
ap.add_argument("--first-step", type=int, default=0); ap.add_argument("--last-step", type=int, default=-1,
                help="oracle steps [first, last) of the chain in this call (a float64 run does not fit one 20-minute GPU-box call): the state after `last` goes to --state-out")
ap.add_argument("--state-in", default=""); ap.add_argument("--state-out", default="")
a = ap.parse_args()
torch.set_num_threads(a.threads)
dev = torch.device("cuda", 0)
B, S, T = 1, a.size, a.steps
sd = synth.make_state_dict(noisediff_param_spec(a.dim), 0)
cond = synth.make_condition(B, S, seed=1)
x_T = synth.make_noise(2, "x_T", B, 4, S)
steps = torch.stack([synth.make_noise(2, f"noise.{i}", B, 4, S) for i in range(T - 1)])

trajs = {}
for form in ("fp32",):
    net = NoiseDiffNet(SimpleNamespace(dim=a.dim, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False))
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    gd = GaussianDiffusion(net, image_size=S, timesteps=T, beta_schedule="sigmoid2", objective="pred_v").to(dev)
    t0 = time.time()
    with torch.inference_mode():
        trajs[form] = gd.sample(batch_size=B, condition={k: v.to(dev) for k, v in cond.items()}, return_all_timesteps=True,
                                noise={"x_T": x_T, "steps": steps}).cpu()            # (B, T+1, C, H, W)
    print(f"HIP sampler ({form}): {time.time() - t0:.1f} s", flush=True)
    del gd, net

odt = torch.float64 if a.oracle_dtype == "f64" else torch.float32
torch.set_default_dtype(odt)
conv = lambda d: {k: (v.to(odt) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()}
buf = O.schedule_buffers("sigmoid2", T, "pred_v")
buf = conv(buf) if isinstance(buf, dict) else buf
sd_o, cond_o = conv(sd), conv(cond)
errs, t0 = {f: {} for f in trajs}, time.time()
state = {"k": 0}

def on_step(t, img, out):
    k = state["k"]                                    # img is the oracle's x_t before step k (k = 0: x_T)
    ref = img.double().numpy()
    for f, traj in trajs.items():
        errs[f][k] = float(np.max(np.abs(traj[:, k].double().numpy() - ref)) / max(1.0, float(np.max(np.abs(ref)))))
    if k % 50 == 0:
        print(f"step {k:4d} (t={t:3d}): rel err of x_t " + ", ".join(f"{f} {errs[f][k]:.3e}" for f in trajs) + f"   [{time.time() - t0:.0f} s]", flush=True)
    state["k"] = k + 1

k0, k1 = a.first_step, (T if a.last_step < 0 else min(a.last_step, T))
if a.state_in:
    stt = torch.load(a.state_in)
    assert stt["step"] == k0, (stt["step"], k0)
    img, draw = stt["img"].to(odt), stt["draw"]
else:
    assert k0 == 0
    img, draw = x_T.to(odt), 0
state["k"] = k0
with torch.no_grad():                                  # the loop of oracle.p_sample_loop (:414-438), resumable
    for k in range(k0, k1):
        t = T - 1 - k
        tt = torch.full((B,), t, dtype=torch.long)
        out = O.noisediff_forward(sd_o, img, tt, cond_o)
        on_step(t, img, out)
        _, x0 = O.predict_x0_eps(buf, "pred_v", img, t, out, clip=False)
        x0 = x0.clamp(-1.0, 1.0)
        mean = O._coef(buf, "posterior_mean_coef1", t) * x0 + O._coef(buf, "posterior_mean_coef2", t) * img
        if t > 0:
            img = mean + (0.5 * O._coef(buf, "posterior_log_variance_clipped", t)).exp() * steps[draw].to(odt)
            draw += 1
        else:
            img = mean + (0.5 * O._coef(buf, "posterior_log_variance_clipped", t)).exp() * 0.0
if a.state_out:
    torch.save({"step": k1, "img": img.double(), "draw": draw}, a.state_out)
ref = img.double()
res = {"config": f"d={a.dim}, {S}x{S}x4, {T}-step DDPM, B=1, explicit noise", "oracle_arithmetic": a.oracle_dtype, "oracle_steps": [k0, k1], "tolerance": 1e-3,
       "oracle_seconds": time.time() - t0, "forms": {}}
for f, traj in trajs.items():
    final = float(np.max(np.abs(traj[:, k1].double().numpy() - ref.numpy())) / max(1.0, float(np.max(np.abs(ref.numpy())))))      # x after step k1 (k1 = T: x_0)
    res["forms"][f] = {"final_rel_err": final, "max_rel_err_over_trajectory": max(errs[f].values()),
                       "rel_err_every_100_steps": {str(k): errs[f][k] for k in sorted(errs[f]) if k % 100 == 0}}
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(REPO, "gpurun_out", f"parity_full_length{'_f64_oracle' if a.oracle_dtype == 'f64' else ''}{f'_{k0}_{k1}' if (k0, k1) != (0, T) else ''}.json"), "w"), indent=1)
print(json.dumps(res["forms"], indent=1))
