"""``NoiseDiffNet``: the arch plug-in of the reference, backed by the HIP engine.

Drop-in contract (models/modules.py:20-41, models/archs/Diffusion_arch.py:447-646):
constructed as ``NoiseDiffNet(args)`` from the reference's argparse namespace; an ``nn.Module``
whose state-dict has the reference's 416 names/shapes (so ``DiffusionNet_ckpt.pth`` loads with
``strict=True`` and EMA deep-copies / DataParallel wrapping work: ``GaussianDiffusion.sample`` shards the batch over the
wrapper's ``device_ids`` itself, and a replica made by ``nn.DataParallel.forward`` uses the owning module's per-device engine); attributes ``channels``,
``out_dim``, ``self_condition``, ``random_or_learned_sinusoidal_cond``, ``downsample_factor``;
``forward(x, time, condition)`` with NCHW tensors and the ``clean_img`` / ``position`` /
``iso_ratio_idx`` condition dict.

Without autograd the forward pass runs ONLY on the fused HIP engine (engine.py); with autograd (training: ``p_losses``) the
NoiseDiffNet graph runs on the differentiable HIP operators of train.py.  There is no CPU or eager-PyTorch fallback: a CPU
tensor or a missing library raises.
"""
from __future__ import annotations

import math
import threading
from typing import Dict, Optional

import torch
from torch import nn

from . import _lib as L
from .spec import arch_param_spec, arch_traits, attention_param_spec, normalize_stage_attn, stage_attention_param_spec


class _Node(nn.Module):
    """Anonymous container used to reproduce the reference's parameter names."""


def _attach(root: nn.Module, dotted: str, param: nn.Parameter) -> None:
    *path, leaf = dotted.split(".")
    m = root
    for part in path:
        if part not in m._modules:
            m.add_module(part, _Node())
        m = m._modules[part]
    m.register_parameter(leaf, param)


def _init(p, gen: Optional[torch.Generator] = None) -> torch.Tensor:
    """PyTorch's default layer init -- the reference never calls init_weights (modules.py:82)."""
    t = torch.empty(p.shape, dtype=torch.float32)
    if p.init == "uniform_fan_in":
        b = 1.0 / math.sqrt(p.fan_in)
        return t.uniform_(-b, b, generator=gen)
    if p.init == "normal":
        return t.normal_(0.0, float(getattr(p, "std", 1.0)), generator=gen)
    return t.fill_(1.0 if p.init == "ones" else 0.0)


class NoiseDiffNet(nn.Module):
    ARCH = "NoiseDiffNet"          # subclasses below: the UNet_PosEmbV2* ablation nets of models/archs/others_arch.py

    def __init__(self, args, mid_attn: Optional[bool] = None, stage_attn=None):
        super().__init__()
        self.dim = int(args.dim)
        if self.dim % 8 or self.dim < 16:
            raise ValueError(f"dim={self.dim}: GroupNorm(8, dim) and the HIP tilings need a multiple of 8, >= 16")
        self.channels = int(args.inp_dim)                       # :458,475
        if self.channels != 4:
            raise ValueError("inp_dim must be 4 (packed RGGB RAW); the 7x7 stem and the sampler kernels assume it")
        self.out_dim = self.channels                            # :550-551
        self.self_condition = args.self_condition               # :469
        self.normalize_condition = args.normalize_condition     # :470
        self.random_or_learned_sinusoidal_cond = False          # :493 (both flags are hard-wired False)
        # BASELINE config 4 extension: Attention between the mid blocks (computed but never wired at :467-468,518)
        self.has_mid_attn = bool(getattr(args, "mid_attn", False) if mid_attn is None else mid_attn)
        if arch_traits(self.ARCH).cond_branch and int(getattr(args, "cond_dim", 4)) != 4:
            raise ValueError("cond_dim must be 4 (the clean image is packed RGGB RAW like the input; 7x7 stem kernel)")
        spec = list(arch_param_spec(self.ARCH, self.dim, self.channels))
        if self.has_mid_attn:
            spec += attention_param_spec("mid_attn", 8 * self.dim)
        # SURVEY 8f-3, second half: upstream's per-stage attention (the classes the reference defines, computes the flags for and drops,
        # Diffusion_arch.py:467-468,509-518): ``args.stage_attn = True`` wires LinearAttention x3 + Attention as the reference's full_attn says
        self.stage_attn = normalize_stage_attn(getattr(args, "stage_attn", None) if stage_attn is None else stage_attn)
        if self.stage_attn:
            if self.ARCH != "NoiseDiffNet":
                raise ValueError("stage_attn is wired for NoiseDiffNet only")
            spec += stage_attention_param_spec(self.dim, self.stage_attn)
        for p in spec:
            _attach(self, p.name, nn.Parameter(_init(p)))
        self._engines: Dict[int, object] = {}
        self._engine_sig: Dict[int, tuple] = {}
        self._lock = threading.Lock()
        self._dp_root: Optional["NoiseDiffNet"] = None          # set on nn.DataParallel replicas: the module that owns the parameters

    @property
    def downsample_factor(self) -> int:                         # :573-575
        return 8

    # copies (EMA's deepcopy, pickling, DataParallel replicas) never share device engines
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_engines"], state["_engine_sig"], state["_lock"] = {}, {}, None
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self._lock = threading.Lock()
        self._dp_root = None

    def _replicate_for_data_parallel(self):
        """nn.DataParallel.forward on several devices (models/modules.py:81): a replica shares ``__dict__`` entries with the module it
        was made from but has no parameters of its own (``replicate`` hands it broadcast copies as plain attributes).  It keeps a
        reference to the owning module and uses ITS engines -- one per device, filled from the owner's parameters (device to device
        from the first engine), so the per-step weight broadcast of the reference's DataParallel never feeds the kernels."""
        replica = super()._replicate_for_data_parallel()
        replica._dp_root = self._dp_root or self
        return replica

    # ------------------------------------------------------------------ engine management
    def _signature(self) -> tuple:
        def version(p):
            try:
                return p._version
            except RuntimeError:                                 # parameters created under torch.inference_mode() have no version counter
                return -1
        return tuple((p.data_ptr(), version(p)) for p in self.parameters())

    def hip_engine(self, device: torch.device, like: Optional[torch.device] = None):
        """The packed-weight engine for ``device`` (rebuilt when parameters changed).  ``like``: another device whose engine is
        current -- the arena is then copied from it device to device instead of being packed again from the state dict."""
        from .engine import Engine
        if self._dp_root is not None:                           # a DataParallel replica: the owner's engines (its parameters are the source)
            return self._dp_root.hip_engine(device, like)
        if device.type != "cuda":
            raise L.HipError(f"{self.ARCH} runs on the HIP library only; tensor is on {device} and there is no CPU path")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        device = torch.device("cuda", idx)
        with self._lock:
            sig = self._signature()
            eng = self._engines.get(idx)
            if eng is None:
                eng = Engine(self.dim, device, mid_attn=self.has_mid_attn, inp_dim=self.channels, arch=self.ARCH, stage_attn=self.stage_attn)
                self._engines[idx] = eng
                self._engine_sig[idx] = None
            if self._engine_sig[idx] != sig:
                src = None
                if like is not None and like.index != idx:
                    src = self._engines.get(like.index)
                    if src is not None and self._engine_sig.get(like.index) != sig:
                        src = None
                if src is None:
                    src = next((e for i, e in self._engines.items() if i != idx and self._engine_sig.get(i) == sig), None)
                if src is not None:
                    eng.copy_from(src)
                else:
                    eng.load_state_dict({k: v for k, v in self.state_dict().items()})
                self._engine_sig[idx] = sig
            return eng

    def adopt_engine(self, eng) -> None:
        """Use an engine whose arena was filled elsewhere (e.g. by the one RCCL broadcast)."""
        with self._lock:
            self._engines[eng.device.index] = eng
            self._engine_sig[eng.device.index] = self._signature()

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor, time: torch.Tensor, condition=None) -> torch.Tensor:
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return self._forward_autograd(x, time, condition)
        assert all(d % self.downsample_factor == 0 for d in x.shape[-2:]), \
            f"your input dimensions {tuple(x.shape[-2:])} need to be divisible by {self.downsample_factor}, given the unet"
        B, Cc, H, W = x.shape
        plan = self.hip_engine(x.device).plan(B, H, W)
        with plan.lock:            # nn.DataParallel.forward calls replicas from several threads: one device's plan serves one call at a time
            plan.set_condition(condition)
            return plan.forward(x, time)


    def _forward_autograd(self, x: torch.Tensor, time: torch.Tensor, condition) -> torch.Tensor:
        """The training entry (models/denoising_diffusion_pytorch.py:481-531 drives ``model(x, t, condition)`` with autograd on,
        models/trainer_diffusion.py:179-190): the same module, the same parameters, on the differentiable operators of train.py --
        3x3 convolutions, GroupNorm (+ modulation + SiLU) and LayerNorm forward and backward and the Linear / 1x1 weight gradients on
        the HIP library -- over the reference graph of trainable.py.  The fused sampling engine has no backward, so this path does
        not go through it; like it, it has no CPU fallback."""
        if x.device.type != "cuda":
            raise L.HipError(f"{self.ARCH} runs on the HIP library only; tensor is on {x.device} and there is no CPU path")
        L.load()
        from .trainable import _Ops, _forward
        # (the ablation nets, the mid-block Attention and the per-stage attention wiring train the same way -- others_arch.py:364-985 go through the same p_losses)
        return _forward(_Ops(self._parameter_table(), True), x, time, condition, arch=self.ARCH, mid_attn=self.has_mid_attn, stage_attn=self.stage_attn)

    def _parameter_table(self) -> Dict[str, torch.Tensor]:
        """name -> tensor for the differentiable forward.  An nn.DataParallel replica holds no Parameters: ``replicate`` leaves the
        broadcast copies (autograd-connected to the owner's parameters, on the replica's device) in ``_former_parameters``."""
        if self._dp_root is None:
            return dict(self.named_parameters())
        table = {}
        for prefix, m in self.named_modules():
            for k, v in getattr(m, "_former_parameters", {}).items():
                table[f"{prefix}.{k}" if prefix else k] = v
        return table


class UNet_PosEmbV2(NoiseDiffNet):
    """models/archs/others_arch.py:364-537: no shot branch, no ISO attention; clean image through cond_init_conv ->
    cond_res_block1 -> cond_concat_conv; condition dict with ``clean_img`` and ``position``."""
    ARCH = "UNet_PosEmbV2"


class UNet_PosEmbV2_NoPosition(NoiseDiffNet):
    """models/archs/others_arch.py:540-707: as UNet_PosEmbV2 without the positional inputs (pos_block1/2 are plain
    ResnetBlocks); ``condition`` is the clean image tensor itself (:658) -- a dict with ``clean_img`` is accepted too."""
    ARCH = "UNet_PosEmbV2_NoPosition"


class UNet_PosEmbV2_CameraCond(NoiseDiffNet):
    """models/archs/others_arch.py:796-985: UNet_PosEmbV2 plus the ISO-conditioned AttnBlock after every stage."""
    ARCH = "UNet_PosEmbV2_CameraCond"
