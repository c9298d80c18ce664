"""``GaussianDiffusion``: the reference's diffusion wrapper with the sampling loop on HIP.

Same constructor, buffers, properties, ``.sample()`` and ``.forward()`` signatures as
models/denoising_diffusion_pytorch.py:167-542, so ``Trainer.__init__`` / ``Trainer.test`` / ``Trainer.train``
(models/trainer_diffusion.py:77-86,286-294,179-181) work unchanged.  The training entry points
(``q_sample`` / ``p_losses`` / ``forward``) are plain differentiable PyTorch around ``self.model``.  ``sample`` accepts two extra
keyword-only arguments for reproducibility (``noise=`` explicit draws for parity, ``seed=``
for the device Philox stream); the reference call signature is a strict subset.

The loop itself is one captured hipGraph per step: [write t] -> time MLP -> U-Net -> fused
x0 / clamp / posterior / noise update -> advance; T replays, no host<->device traffic inside.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import threading
from collections import namedtuple
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib as L


BUFFER_NAMES = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
                "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
                "loss_weight"]

ModelPrediction = namedtuple("ModelPrediction", ["pred_noise", "pred_x_start"])       # as the reference's (:36)

_SIGMOID = {"sigmoid1": (-3, 3, 0.5), "sigmoid2": (-7, 3, 0.7), "sigmoid3": (-10, 3, 0.7)}


def make_betas(name: str, timesteps: int, **kw) -> torch.Tensor:
    """fp64 beta schedules (denoising_diffusion_pytorch.py:96-164); unknown names raise as at :218
    (the CLI default 'sigmoid' is NOT a valid name in the reference and stays invalid here)."""
    f64 = torch.float64
    if name == "linear":
        scale = 1000 / timesteps
        return torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=f64)
    grid = torch.linspace(0, timesteps, timesteps + 1, dtype=f64) / timesteps
    if name == "cosine":
        s = kw.get("s", 0.008)
        bar = torch.cos((grid + s) / (1 + s) * torch.pi * 0.5) ** 2
    elif name in _SIGMOID:
        start, end, tau = _SIGMOID[name]
        start, end, tau = kw.get("start", start), kw.get("end", end), kw.get("tau", tau)
        lo, hi = torch.tensor(start / tau).sigmoid(), torch.tensor(end / tau).sigmoid()   # fp32 scalars, as upstream
        bar = (hi - ((grid * (end - start) + start) / tau).sigmoid()) / (hi - lo)
    else:
        raise ValueError(f"unknown beta schedule {name}")
    bar = bar / bar[0]
    return torch.clip(1 - bar[1:] / bar[:-1], 0, 0.999)


def make_buffers(betas: torch.Tensor, objective: str) -> Dict[str, torch.Tensor]:
    """The 13 schedule buffers (:222-286), fp64 math, stored fp32."""
    a = 1.0 - betas
    ac = torch.cumprod(a, dim=0)
    ac_prev = F.pad(ac[:-1], (1, 0), value=1.0)
    pv = betas * (1.0 - ac_prev) / (1.0 - ac)
    snr = ac / (1 - ac)
    lw = {"pred_noise": snr / snr, "pred_x0": snr.clone(), "pred_v": snr / (snr + 1)}[objective]
    vals = [betas, ac, ac_prev, ac.sqrt(), (1.0 - ac).sqrt(), (1.0 - ac).log(), (1.0 / ac).sqrt(), (1.0 / ac - 1).sqrt(),
            pv, pv.clamp(min=1e-20).log(), betas * ac_prev.sqrt() / (1.0 - ac), (1.0 - ac_prev) * a.sqrt() / (1.0 - ac), lw]
    return {n: v.to(torch.float32) for n, v in zip(BUFFER_NAMES, vals)}


def _rows(v, lo: int, hi: int, B: int):
    """Batch rows [lo, hi) of a tensor, or of every tensor of a condition dict (whose leading dimension is the batch)."""
    if v is None:
        return None
    if isinstance(v, dict):
        return {k: _rows(t, lo, hi, B) for k, t in v.items()}
    if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == B:
        return v[lo:hi]
    return v


def _unwrap(model: nn.Module) -> nn.Module:
    return model.module if isinstance(model, (nn.DataParallel, nn.parallel.DistributedDataParallel)) else model


class GaussianDiffusion(nn.Module):
    def __init__(self, model, *, image_size, timesteps=1000, sampling_timesteps=None, objective="pred_v",
                 beta_schedule="sigmoid", schedule_fn_kwargs=dict(), ddim_sampling_eta=0., auto_normalize=False,
                 offset_noise_strength=0., min_snr_gamma=5):
        super().__init__()
        net = _unwrap(model)       # the reference crashes on a bare module (:189); both forms are accepted here
        assert not (type(self) == GaussianDiffusion and net.channels != net.out_dim)
        assert not net.random_or_learned_sinusoidal_cond
        self.model = model
        self.channels = net.channels
        self.self_condition = net.self_condition
        self.image_size = image_size
        self.objective = objective
        assert objective in {"pred_noise", "pred_x0", "pred_v"}, \
            "objective must be either pred_noise (predict noise) or pred_x0 (predict image start) or pred_v (predict v)"
        betas = make_betas(beta_schedule, timesteps, **schedule_fn_kwargs)
        self.num_timesteps = int(betas.shape[0])
        self.sampling_timesteps = sampling_timesteps if sampling_timesteps is not None else self.num_timesteps
        assert self.sampling_timesteps <= self.num_timesteps
        self.is_ddim_sampling = self.sampling_timesteps < self.num_timesteps
        self.ddim_sampling_eta = ddim_sampling_eta
        for name, val in make_buffers(betas, objective).items():
            self.register_buffer(name, val)
        self.offset_noise_strength = offset_noise_strength
        self.auto_normalize = auto_normalize
        # rank shard bookkeeping for the device noise stream (see shard.py): global index of sample 0
        self.sample_offset = 0
        self._loop_cache: Dict[tuple, "_Loop"] = {}
        # p_sample_loop / ddim_sample switch THIS wrapper's sampler for one call (_as_sampler): a lock per instance, so that wrappers on other devices or
        # in other threads never wait for each other
        self._sampler_switch = threading.RLock()

    # ------------------------------------------------------------------ reference-compatible helpers
    def normalize(self, img):
        return img * 2 - 1 if self.auto_normalize else img

    def unnormalize(self, t):
        return (t + 1) * 0.5 if self.auto_normalize else t

    @property
    def device(self):
        return self.betas.device

    def ddim_time_pairs(self):
        times = torch.linspace(-1, self.num_timesteps - 1, steps=self.sampling_timesteps + 1)   # fp32, as :409
        times = list(reversed(times.int().tolist()))
        return list(zip(times[:-1], times[1:]))

    # ------------------------------------------------------------------ per-step coefficient tables
    def _tables(self):
        """(t_cur, t_next, coef[n,8]) on the CPU, built from the fp32 buffers exactly as the
        reference combines them per step (:322-329,:372 for DDPM; :427-431 for DDIM)."""
        cpu = {n: getattr(self, n).detach().cpu() for n in BUFFER_NAMES}
        rows: List[torch.Tensor] = []
        if not self.is_ddim_sampling:
            ts = list(reversed(range(self.num_timesteps)))
            nxt = [t - 1 for t in ts]
            for t in ts:
                rows.append(torch.stack([cpu["sqrt_alphas_cumprod"][t], cpu["sqrt_one_minus_alphas_cumprod"][t],
                                         cpu["sqrt_recip_alphas_cumprod"][t], cpu["sqrt_recipm1_alphas_cumprod"][t],
                                         cpu["posterior_mean_coef1"][t], cpu["posterior_mean_coef2"][t],
                                         (0.5 * cpu["posterior_log_variance_clipped"][t]).exp(),
                                         torch.tensor(1.0 if t > 0 else 0.0)]))
        else:
            pairs = self.ddim_time_pairs()
            ts, nxt = [p[0] for p in pairs], [p[1] for p in pairs]
            ac, eta = cpu["alphas_cumprod"], self.ddim_sampling_eta
            for t, tn in pairs:
                head = [cpu["sqrt_alphas_cumprod"][t], cpu["sqrt_one_minus_alphas_cumprod"][t],
                        cpu["sqrt_recip_alphas_cumprod"][t], cpu["sqrt_recipm1_alphas_cumprod"][t]]
                if tn < 0:
                    tail = [torch.tensor(0.0), torch.tensor(0.0), torch.tensor(0.0), torch.tensor(1.0)]
                else:
                    a, an = ac[t], ac[tn]
                    sigma = eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
                    c = (1 - an - sigma ** 2).sqrt()
                    tail = [an.sqrt(), c, sigma, torch.tensor(0.0)]
                rows.append(torch.stack(head + tail))
        return (torch.tensor(ts, dtype=torch.int32), torch.tensor(nxt, dtype=torch.int32),
                torch.stack(rows).to(torch.float32).contiguous())

    # ------------------------------------------------------------------ sampling
    _MAX_LOOPS = 4       # captured step graphs kept per wrapper (one per (plan, sampler settings))
    _SHARD_BLOCK = 8     # one process, several devices: steps queued per device and turn (see sample())

    def _loop_for(self, plan) -> "_Loop":
        key = (id(plan), self.is_ddim_sampling, self.objective, self.sampling_timesteps, float(self.ddim_sampling_eta))
        loop = self._loop_cache.pop(key, None)
        if loop is None:
            loop = _Loop(self, plan)
            while len(self._loop_cache) >= max(self._MAX_LOOPS, 2 * len(self._sampling_devices())):   # (one loop per device shard stays resident)
                self._loop_cache.pop(next(iter(self._loop_cache))).destroy()     # oldest first; frees its hipGraphExec
        self._loop_cache[key] = loop                                            # most recently used last
        return loop

    @torch.inference_mode()
    def sample(self, batch_size=16, condition=None, return_all_timesteps=False, preset_mean=None, *,
               noise: Optional[Dict[str, torch.Tensor]] = None, seed: Optional[int] = None):
        """(B, C, H, W) fp32 samples, or (B, T+1, C, H, W) with ``return_all_timesteps`` (:446-451).

        noise={'x_T': (B,C,H,W), 'steps': (n_draws,B,C,H,W)} injects the exact draws the reference would take
        from torch.randn / randn_like (parity mode).  Otherwise x_T and the per-step noise come from the
        device Philox stream keyed by (seed, global sample index, step); seed=None draws one from torch's
        default generator, so torch.manual_seed() makes runs repeatable.  The seed lives in device memory, so
        the captured step graph is reused by every call (no re-capture per seed).
        """
        net = _unwrap(self.model)
        if not hasattr(net, "hip_engine"):
            raise TypeError("noisediff_amd.GaussianDiffusion.sample drives a noisediff_amd network (HIP engine); got "
                            f"{type(net).__name__}.  Other nn.Modules are accepted for the training entry points only.")
        B, S = int(batch_size), self.image_size
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        x_T = None
        if noise is not None and "x_T" in noise:
            x_T = noise["x_T"]
        if preset_mean is not None and not self.is_ddim_sampling:   # ddim_sample accepts but ignores preset_mean (:405,413)
            x_T = preset_mean
        step_noise = None if noise is None else noise.get("steps")
        devices = self._sampling_devices()
        if len(devices) == 1:
            plan = net.hip_engine(devices[0]).plan(B, S, S)
            plan.set_condition(condition)
            ret = self._loop_for(plan).run(x_T=x_T, step_noise=step_noise, seed=seed, first_sample=int(self.sample_offset),
                                           return_all=return_all_timesteps)
            return self.unnormalize(ret)
        # ---- nn.DataParallel(net, device_ids=[...]) with several devices: the reference's --gpu_ids entry (models/modules.py:73-83).
        # The reference replicates the weights and scatters / gathers the batch on EVERY step (:332); here the batch rows are split
        # once, every device gets its own engine (arena copied device to device from the first), plan and captured step graph, and
        # the host thread queues each device's graph replays round-robin, a block of steps at a time: no per-step traffic between the devices.  Row i of the batch draws the
        # same Philox noise (keyed by its global index) and runs the same kernels whatever the number of devices.
        from .shard import shard_bounds
        shards = []
        for k, dev in enumerate(devices):
            lo, hi = shard_bounds(B, k, len(devices))
            if hi == lo:
                continue
            plan = net.hip_engine(dev, like=devices[0]).plan(hi - lo, S, S, tag=k)
            plan.set_condition(_rows(condition, lo, hi, B))
            loop = self._loop_for(plan)
            loop.start(_rows(x_T, lo, hi, B), None if step_noise is None else step_noise[:, lo:hi], seed, int(self.sample_offset) + lo)
            shards.append((loop, plan))
        n_steps = shards[0][0].n_steps
        out_dev = self.device
        gather = lambda: torch.cat([p.read_nchw(p.x).to(out_dev) for _, p in shards], dim=0)
        if not return_all_timesteps:
            # the shards advance round-robin in blocks of _SHARD_BLOCK steps: every device's stream always holds a few queued graph replays (no waiting
            # for this thread between steps), and a hipGraphLaunch that blocks on a full queue delays the other devices by at most one block -- queueing
            # one device's whole chain first (r5) leaves the others idle for as long as that queue stays full.  The bits do not depend on the block.
            done = 0
            while done < n_steps:
                n = min(self._SHARD_BLOCK, n_steps - done)
                for loop, _p in shards:
                    loop.advance(n)
                done += n
            return self.unnormalize(gather())
        frames = [gather()]
        for _ in range(n_steps):
            for loop, _p in shards:
                loop.advance(1)
            frames.append(gather())
        return self.unnormalize(torch.stack(frames, dim=1))

    def _sampling_devices(self) -> List[torch.device]:
        """The devices sample() shards the batch over: the wrapper's own device, or -- when the model is wrapped in
        nn.DataParallel with several device_ids, as ``define_G`` does for ``--gpu_ids 0,1,...`` -- those devices."""
        if isinstance(self.model, nn.DataParallel) and len(self.model.device_ids) > 1:
            if self.device.type != "cuda":
                raise L.HipError(f"noisediff_amd runs on MI355X only: the diffusion wrapper is on {self.device}")
            return [torch.device("cuda", int(i)) for i in self.model.device_ids]
        return [self.device]

    # copies (EMA's deepcopy of the trainer, pickling) never carry device loops: they hold ctypes handles
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_loop_cache"] = {}
        state.pop("_sampler_switch", None)          # (a lock is neither copied nor pickled: __setstate__ makes a new one)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self._sampler_switch = threading.RLock()

    def release_graphs(self) -> None:
        """Destroy every captured step graph of this wrapper (they are rebuilt on demand)."""
        for loop in self._loop_cache.values():
            loop.destroy()
        self._loop_cache = {}

    # ------------------------------------------------------------------ training entry points: plain differentiable PyTorch
    # (models/denoising_diffusion_pytorch.py:473-542).  No HIP kernels here: `self.model` may be ANY differentiable
    # nn.Module with the arch plug-in signature model(x, t, condition) -- e.g. the reference NoiseDiffNet while its
    # weights are trained, which then load into noisediff_amd.NoiseDiffNet for sampling (same state dict).
    @staticmethod
    def _at(table: torch.Tensor, t: torch.Tensor, ndim: int) -> torch.Tensor:
        """table[t] shaped (B, 1, ..., 1) for broadcasting against an ndim-dimensional batch (`extract`, :91-94)."""
        return table.gather(-1, t).reshape(t.shape[0], *((1,) * (ndim - 1)))

    # ------------------------------------------------------------------ the reference's per-step public methods (:298-373, :453-471)
    # sample() runs the whole chain fused on the device (one step kernel + one captured graph per step); these are the same quantities one
    # call at a time, for callers that drive the chain themselves: the network call goes to self.model (the HIP engine under no_grad), the
    # elementwise part is plain torch on the caller's tensors.
    def predict_start_from_noise(self, x_t, t, noise):          # :298-302
        return self._at(self.sqrt_recip_alphas_cumprod, t, x_t.ndim) * x_t - self._at(self.sqrt_recipm1_alphas_cumprod, t, x_t.ndim) * noise

    def predict_noise_from_start(self, x_t, t, x0):             # :304-308
        return (self._at(self.sqrt_recip_alphas_cumprod, t, x_t.ndim) * x_t - x0) / self._at(self.sqrt_recipm1_alphas_cumprod, t, x_t.ndim)

    def predict_start_from_v(self, x_t, t, v):                  # :316-320
        return self._at(self.sqrt_alphas_cumprod, t, x_t.ndim) * x_t - self._at(self.sqrt_one_minus_alphas_cumprod, t, x_t.ndim) * v

    def q_posterior(self, x_start, x_t, t):                     # :322-329 -> (mean, variance, clipped log variance)
        mean = self._at(self.posterior_mean_coef1, t, x_t.ndim) * x_start + self._at(self.posterior_mean_coef2, t, x_t.ndim) * x_t
        return mean, self._at(self.posterior_variance, t, x_t.ndim), self._at(self.posterior_log_variance_clipped, t, x_t.ndim)

    def model_predictions(self, x, t, condition=None, clip_x_start=False, rederive_pred_noise=False):      # :331-354
        out = self.model(x, t, condition)
        clip = (lambda v: v.clamp(-1., 1.)) if clip_x_start else (lambda v: v)
        if self.objective == "pred_noise":
            pred_noise = out
            x_start = clip(self.predict_start_from_noise(x, t, pred_noise))
            if clip_x_start and rederive_pred_noise:
                pred_noise = self.predict_noise_from_start(x, t, x_start)
        elif self.objective == "pred_x0":
            x_start = clip(out)
            pred_noise = self.predict_noise_from_start(x, t, x_start)
        else:                                                   # pred_v
            x_start = clip(self.predict_start_from_v(x, t, out))
            pred_noise = self.predict_noise_from_start(x, t, x_start)
        return ModelPrediction(pred_noise, x_start)

    def p_mean_variance(self, x, t, condition=None, clip_denoised=True):                                    # :356-364
        x_start = self.model_predictions(x, t, condition).pred_x_start
        if clip_denoised:
            x_start = x_start.clamp(-1., 1.)
        mean, var, logvar = self.q_posterior(x_start=x_start, x_t=x, t=t)
        return mean, var, logvar, x_start

    @torch.inference_mode()
    def p_sample(self, x, t: int, condition=None, noise=None):                                              # :366-373 -> (x_{t-1}, x_start)
        """One reverse step at integer timestep ``t``; ``noise`` (optional) replaces the reference's randn_like draw."""
        times = torch.full((x.shape[0],), int(t), device=x.device, dtype=torch.long)
        mean, _, logvar, x_start = self.p_mean_variance(x=x, t=times, condition=condition, clip_denoised=True)
        if t > 0:
            z = torch.randn_like(x) if noise is None else noise.to(x)
            return mean + (0.5 * logvar).exp() * z, x_start
        return mean, x_start

    @contextlib.contextmanager
    def _as_sampler(self, ddim: bool):
        """Run sample() as the named sampler whatever the wrapper was configured with (p_sample_loop / ddim_sample are public in the reference)."""
        with self._sampler_switch:                  # (the attribute is read by sample() and its loops: another thread sampling on THIS wrapper waits here)
            saved = self.is_ddim_sampling
            self.is_ddim_sampling = ddim
            try:
                yield
            finally:
                self.is_ddim_sampling = saved

    def _check_shape(self, shape):
        if tuple(shape[1:]) != (self.channels, self.image_size, self.image_size):
            raise ValueError(f"shape {tuple(shape)} does not match (B, {self.channels}, {self.image_size}, {self.image_size})")

    @torch.inference_mode()
    def p_sample_loop(self, shape, condition=None, return_all_timesteps=False, preset_mean=None, **kw):           # :375-402
        self._check_shape(shape)
        with self._as_sampler(False):
            return self.sample(batch_size=shape[0], condition=condition, return_all_timesteps=return_all_timesteps, preset_mean=preset_mean, **kw)

    @torch.inference_mode()
    def ddim_sample(self, shape, condition=None, return_all_timesteps=False, preset_mean=None, **kw):             # :404-444 (preset_mean accepted, ignored)
        self._check_shape(shape)
        with self._as_sampler(True):
            return self.sample(batch_size=shape[0], condition=condition, return_all_timesteps=return_all_timesteps, preset_mean=preset_mean, **kw)

    @torch.inference_mode()
    def interpolate(self, x1, x2, t=None, lam=0.5, condition=None):                                         # :453-471
        """q_sample both images to timestep t, mix them, walk back with p_sample.  The reference hands `self_cond` (None) to the network's
        `condition` slot, which NoiseDiffNet cannot run with; ``condition`` is this build's way to pass the conditioning through."""
        t = self.num_timesteps - 1 if t is None else int(t)
        assert x1.shape == x2.shape
        tb = torch.full((x1.shape[0],), t, device=x1.device, dtype=torch.long)
        img = (1 - lam) * self.q_sample(x1, tb) + lam * self.q_sample(x2, tb)
        for i in reversed(range(0, t)):
            img, _ = self.p_sample(img, i, condition)
        return img

    def predict_v(self, x_start, t, noise):                     # :310-314
        return (self._at(self.sqrt_alphas_cumprod, t, x_start.ndim) * noise
                - self._at(self.sqrt_one_minus_alphas_cumprod, t, x_start.ndim) * x_start)

    def q_sample(self, x_start, t, noise=None):                 # :473-479: x_t = sqrt(ac) x_0 + sqrt(1 - ac) eps
        if noise is None:
            noise = torch.randn_like(x_start)
        return (self._at(self.sqrt_alphas_cumprod, t, x_start.ndim) * x_start
                + self._at(self.sqrt_one_minus_alphas_cumprod, t, x_start.ndim) * noise)

    def p_losses(self, x_start, t, condition=None, noise=None, offset_noise_strength=None):
        """Per-sample weighted MSE between the model output and the objective's target (:481-531)."""
        if noise is None:
            noise = torch.randn_like(x_start)
        strength = self.offset_noise_strength if offset_noise_strength is None else offset_noise_strength
        if strength > 0.:                                       # offset noise: one draw per (sample, channel), :490-492
            offset = torch.randn(x_start.shape[:2], device=self.device)
            noise = noise + strength * offset[:, :, None, None]
        x_t = self.q_sample(x_start, t, noise)
        model_out = self.model(x_t, t, condition)
        if self.objective == "pred_noise":
            target = noise
        elif self.objective == "pred_x0":
            target = x_start
        elif self.objective == "pred_v":
            target = self.predict_v(x_start, t, noise)
        else:
            raise ValueError(f"unknown objective {self.objective}")
        per_sample = F.mse_loss(model_out, target, reduction="none").flatten(1).mean(dim=1)
        loss = (per_sample * self.loss_weight.gather(-1, t)).mean()
        if self.objective == "pred_x0":                         # the reference adds a mean-intensity term here (:521-525)
            loss = loss + (model_out.mean(dim=(2, 3)) - target.mean(dim=(2, 3))).abs().mean()
        return loss

    def forward(self, img, condition, *args, **kwargs):
        """Training loss for a batch: uniform random timesteps, then p_losses (:534-542)."""
        b, _c, h, w = img.shape
        assert h == self.image_size and w == self.image_size, f"height and width of image must be {self.image_size}"
        t = torch.randint(0, self.num_timesteps, (b,), device=img.device).long()
        return self.p_losses(self.normalize(img), t, condition, *args, **kwargs)


class _Loop:
    """Device-resident sampler state + the captured step graph for one plan."""

    def __init__(self, gd: GaussianDiffusion, plan):
        self.gd, self.plan = gd, plan
        dev = plan.dev
        t_cur, t_next, coef = gd._tables()
        self.n_steps = int(t_cur.numel())
        with torch.cuda.device(dev), torch.inference_mode(False):
            self.t_cur, self.t_next, self.coef = t_cur.to(dev), t_next.to(dev), coef.to(dev)
            self.step = torch.zeros(1, dtype=torch.int32, device=dev)
            self.rng = torch.zeros(2, dtype=torch.int64, device=dev)      # {seed, first_sample}: read by the step kernel
            torch.cuda.synchronize(dev)
        st = L.SamplerState()
        st.step, st.t_cur, st.t_next, st.coef = self.step.data_ptr(), self.t_cur.data_ptr(), self.t_next.data_ptr(), self.coef.data_ptr()
        st.time_out, st.rng, st.n_steps, st.B = plan.time.data_ptr(), self.rng.data_ptr(), self.n_steps, plan.B
        self.state = st
        self.graph = C.c_void_p()
        self.graph_key = None
        self.noise_nhwc = None

    def destroy(self) -> None:
        if getattr(self, "graph", None):
            try:
                self.plan.e.sync()
                L.call("nd_graph_destroy", self.graph)
            finally:
                self.graph, self.graph_key = C.c_void_p(), None

    def __del__(self):
        try:
            self.destroy()
        except Exception:        # interpreter shutdown: the library or the device may already be gone
            pass

    def _step_eager(self, noise_ptr, stride):
        p, e, st = self.plan, self.plan.e, self.plan.e.stream
        L.call("nd_sampler_begin_step", C.byref(self.state), st)
        p.run(p.step_ops)
        fn = "nd_sampler_step_ddim_f32" if self.gd.is_ddim_sampling else "nd_sampler_step_ddpm_f32"
        # seed / first_sample arguments are overridden by state.rng (device memory): nothing call-specific is baked in
        L.call(fn, p.x.data_ptr(), p.model_out.data_ptr(), noise_ptr, stride, C.byref(self.state),
               L.OBJECTIVES[self.gd.objective], C.c_uint64(0), 0, p.B, p.H * p.W, e.inp_dim, st)
        L.call("nd_sampler_advance", C.byref(self.state), st)

    def _ensure_graph(self, noise_ptr, stride):
        key = (noise_ptr, stride)
        if self.graph_key == key:
            return
        st = self.plan.e.stream
        self.destroy()
        with torch.cuda.device(self.plan.dev):
            L.call("nd_graph_begin", st)
            try:
                self._step_eager(noise_ptr, stride)
            finally:
                L.call("nd_graph_end", st, C.byref(self.graph))
        self.graph_key = key

    def start(self, x_T, step_noise, seed, first_sample, use_graph=True):
        """Reset the device loop state, load or draw x_T and capture the step graph if there is none yet."""
        p, e = self.plan, self.plan.e
        B, Cc, H, W = p.B, e.inp_dim, p.H, p.W
        st = e.stream
        self.use_graph = use_graph
        with torch.cuda.device(p.dev):
            e.sync()                                           # a previous run may still read step / rng
            self.step.zero_()
            seed = int(seed) & (2 ** 64 - 1)                   # any 64-bit seed: stored as its two's-complement int64 (the kernel casts back)
            self.rng.copy_(torch.tensor([seed - 2 ** 64 if seed >= 2 ** 63 else seed, int(first_sample)], dtype=torch.int64))
            noise_ptr, stride = None, 0
            if step_noise is not None:
                need = self.n_steps - 1
                if step_noise.shape[0] < need or tuple(step_noise.shape[1:]) != (B, Cc, H, W):
                    raise ValueError(f"noise['steps'] must be (>={need}, {B}, {Cc}, {H}, {W}); got {tuple(step_noise.shape)}")
                nh = step_noise.to(p.dev, torch.float32).permute(0, 1, 3, 4, 2)
                if self.noise_nhwc is not None and self.noise_nhwc.shape == nh.shape:
                    self.noise_nhwc.copy_(nh)                  # same buffer -> same captured graph
                else:
                    with torch.inference_mode(False):
                        self.noise_nhwc = torch.empty(nh.shape, dtype=torch.float32, device=p.dev)
                    self.noise_nhwc.copy_(nh)
                noise_ptr, stride = self.noise_nhwc.data_ptr(), B * Cc * H * W
            torch.cuda.synchronize(p.dev)
            if x_T is not None:
                if tuple(x_T.shape) != (B, Cc, H, W):
                    raise ValueError(f"x_T / preset_mean must be {(B, Cc, H, W)}; got {tuple(x_T.shape)}")
                p.load_x(x_T)
            else:
                L.call("nd_philox_normal_f32", p.x.data_ptr(), C.c_uint64(seed), first_sample, -1, B, H * W, Cc, st)
            self._args = (noise_ptr, stride)
            if use_graph:
                self._ensure_graph(*self._args)

    def advance(self, n: int = 1) -> None:
        """Enqueue n diffusion steps on the library stream (no host synchronisation)."""
        st = self.plan.e.stream
        # sample() drives the shards of several devices from one host thread (nn.DataParallel(device_ids=[...])): a graph or a kernel is
        # launched with ITS device current (nd_graph_launch also sets the stream's device itself)
        with torch.cuda.device(self.plan.dev):
            for _ in range(n):
                if self.use_graph:
                    L.call("nd_graph_launch", self.graph, st)
                else:
                    self._step_eager(*self._args)

    def run(self, x_T, step_noise, seed, first_sample, return_all=False, use_graph=True):
        p = self.plan
        self.start(x_T, step_noise, seed, first_sample, use_graph)
        if not return_all:
            self.advance(self.n_steps)
            return p.read_nchw(p.x)          # synchronises the library stream
        frames = [p.read_nchw(p.x)]
        for _ in range(self.n_steps):
            self.advance(1)
            frames.append(p.read_nchw(p.x))
        return torch.stack(frames, dim=1)
