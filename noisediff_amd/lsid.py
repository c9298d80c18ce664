"""``LSID``: the reference's denoiser arch (models/archs/SID_arch.py:49-175) on the HIP library -- the
consumer of the synthesized noise (BASELINE config 5).

Same plug-in contract as NoiseDiffNet: ``LSID(args)``, reference state-dict names/shapes (strict load),
``forward(x)`` with an NCHW (B, 4, H, W) tensor.  It reuses the sampler's kernels: every ``Conv2d(3x3)`` is
``nd_conv3x3(_wino2)_nhwc_f32`` storing the *pre-activation*; ``LeakyReLU(0.2)`` is applied by the consumer's
prologue (ND_PRO_LEAKY; it commutes with max-pooling, and ND_PRO_LEAKY_SECOND handles
``cat(up(x), skip)`` where only the skip is activated); ``ConvTranspose2d(2, s=2)`` is one pointwise GEMM to
4*C' columns with a pixel-shuffle store that also performs the crop.  Inference only; no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Tuple

import torch
from torch import nn

from . import _lib as L
from .net import _attach, _init
from .spec import LSID_STAGES, lsid_param_spec

import os
WINO4 = os.environ.get("ND_WINO4", "1") != "0"           # A-B knob: 0 = never the F(4x4,3x3) kernel


class LSID(nn.Module):
    def __init__(self, args=None):
        super().__init__()
        self.block_size = 2
        for p in lsid_param_spec():
            _attach(self, p.name, nn.Parameter(_init(p)))
        self._plans: Dict[Tuple[int, int, int, int], "_LsidPlan"] = {}
        self._sig = None

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_plans"], state["_sig"] = {}, None
        return state

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("noisediff_amd.LSID is inference-only; train with the reference's LSID (same state dict)")
        if x.device.type != "cuda":
            raise L.HipError(f"LSID runs on the HIP library only; tensor is on {x.device} and there is no CPU path")
        sig = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if sig != self._sig:
            self._plans, self._sig = {}, sig
        B, Cc, H, W = x.shape
        key = (x.device.index or 0, B, H, W)
        if key not in self._plans:
            self._plans[key] = _LsidPlan(self, x.device, B, H, W)
        return self._plans[key].run(x)


class _LsidPlan:
    """Packed weights + workspace + recorded launches for one input shape."""

    def __del__(self):                      # the plan owns its HIP stream
        s, self.stream = getattr(self, "stream", None), None
        if s:
            try:
                L.call("nd_stream_sync", s)
                L.call("nd_stream_destroy", s)
            except Exception:               # interpreter shutdown: the library or the device may already be gone
                pass

    def __init__(self, net: LSID, dev: torch.device, B: int, H: int, W: int):
        self.lib = L.load()
        self.dev, self.B, self.H, self.W = dev, B, H, W
        self.keep: List[object] = []
        self.ops: List[tuple] = []
        with torch.cuda.device(dev), torch.inference_mode(False):
            s = C.c_void_p()
            L.call("nd_stream_create", C.byref(s))
            self.stream = s
            sd = {k: v.detach().to(dev, torch.float32).contiguous() for k, v in net.state_dict().items()}
            torch.cuda.synchronize(dev)
            self.w: Dict[str, torch.Tensor] = {}
            for name, t in sd.items():
                if name.endswith(".bias"):
                    self.w[name] = t
                elif name.startswith("up"):          # (Cin, Cout, 2, 2) -> rows n = (p1 p2 c'), columns k = cin
                    cin, cout = t.shape[:2]
                    m = t.permute(2, 3, 1, 0).reshape(4 * cout, cin).contiguous()
                    self.w[name] = self._pack_pw(m)
                elif t.shape[-1] == 1:
                    self.w[name] = self._pack_pw(t.reshape(t.shape[0], -1).contiguous())
                else:
                    if t.shape[1] % 8:               # conv1_1: 4 input channels, zero-padded to 8
                        t = torch.cat((t, torch.zeros(t.shape[0], 8 - t.shape[1] % 8, 3, 3, device=dev)), 1).contiguous()
                        torch.cuda.synchronize(dev)
                    self.w[name] = self._pack_conv(t)
            self.x_nchw = torch.empty(B, 4, H, W, device=dev)
            self.out_nchw = torch.empty(B, 4, H, W, device=dev)
            self._record()
            L.call("nd_stream_sync", self.stream)

    # ------------------------------------------------------------------ helpers
    def _f(self, *shape) -> torch.Tensor:
        t = torch.empty(*shape, dtype=torch.float32, device=self.dev)
        self.keep.append(t)
        return t

    def _pack_pw(self, m: torch.Tensor) -> torch.Tensor:
        cout, cin = m.shape
        out = self._f(self.lib.nd_pack_pointwise_weight_floats(cin, cout))
        self.keep.append(m)
        torch.cuda.synchronize(self.dev)
        L.call("nd_pack_pointwise_weight", m.data_ptr(), out.data_ptr(), cin, cout, 0, self.stream)
        return out

    def _pack_conv(self, t: torch.Tensor):
        """(direct, F(2x2,3x3), F(4x4,3x3) or None) packings of one 3x3 weight."""
        cout, cin = t.shape[:2]
        d = self._f(self.lib.nd_pack_conv3x3_weight_floats(cin, cout))
        w = self._f(self.lib.nd_pack_conv3x3_wino_weight_floats(cin, cout))
        self.keep.append(t)
        L.call("nd_pack_conv3x3_weight", t.data_ptr(), d.data_ptr(), cin, cout, self.stream)
        L.call("nd_pack_conv3x3_wino_weight", t.data_ptr(), w.data_ptr(), cin, cout, self.stream)
        w4 = None
        if WINO4 and cin > 16 and cin % 4 == 0 and cout % 4 == 0:
            w4 = self._f(self.lib.nd_pack_conv3x3_wino4_weight_floats(cin, cout))
            L.call("nd_pack_conv3x3_wino4_weight", t.data_ptr(), w4.data_ptr(), cin, cout, self.stream)
        return d, w, w4

    def _add(self, name: str, *args) -> None:
        self.keep.append(args)
        self.ops.append((getattr(self.lib, name), args, name))

    def _src(self, t, t2=None, mode=L.PRO_NONE) -> L.Src:
        s = L.Src()
        s.p0, s.c0, s.ld0 = t.data_ptr(), t.shape[-1], t.shape[-1]
        if t2 is not None:
            s.p1, s.c1, s.ld1 = t2.data_ptr(), t2.shape[-1], t2.shape[-1]
        s.mode = mode
        return s

    def _conv(self, name: str, src: L.Src, cin: int, cout: int, h: int, w: int) -> torch.Tensor:
        out = self._f(self.B, h, w, cout)
        wino = h >= 16 and w >= 16
        d = L.Conv3x3()
        d.src, d.weight, d.bias, d.out = src, self.w[name + ".weight"][1 if wino else 0].data_ptr(), self.w[name + ".bias"].data_ptr(), out.data_ptr()
        d.B, d.H, d.W, d.cin, d.cout, d.ldo = self.B, h, w, cin, cout, cout
        wino2 = wino and (src.c1 == 0 or src.c0 % 32 == 0) and self.B * h * w < (1 << 24) and self.B * h * w * 4 * max(src.ld0, src.ld1) < (1 << 31)
        # F(4x4,3x3) where its kernel takes the layer (same rule as the engine's): the LeakyReLU prologues are applied while the halo is staged
        wino4 = (wino and self.w[name + ".weight"][2] is not None and w >= 32 and (w % 32 == 0 or w >= 96) and w <= 2048
                 and (src.c1 == 0 or src.c0 % 16 == 0) and self.B * h * w + w + 2 < (1 << 24)
                 and (self.B * h * w + w + 2) * 4 * max(src.ld0, src.ld1) < (1 << 30) - (1 << 16))
        if wino4:
            d.weight = self.w[name + ".weight"][2].data_ptr()
        self._add("nd_conv3x3_wino4_nhwc_f32" if wino4 else "nd_conv3x3_wino2_nhwc_f32" if wino2 else
                  "nd_conv3x3_wino_nhwc_f32" if wino else "nd_conv3x3_nhwc_f32", C.byref(d), self.stream)
        self.keep.append(d)
        return out

    # ------------------------------------------------------------------ the network (SID_arch.py:105-175)
    def _record(self) -> None:
        B, H, W = self.B, self.H, self.W
        x8 = self._f(B, H, W, 8)
        self._add("nd_nchw_to_nhwc_pad_f32", self.x_nchw.data_ptr(), x8.data_ptr(), B, 4, H, W, 8, self.stream)
        feats: List[Tuple[torch.Tensor, int, int]] = []
        x, mode, cin, h, w = x8, L.PRO_NONE, 8, H, W
        for i, c in enumerate(LSID_STAGES, start=1):
            a = self._conv(f"conv{i}_1", self._src(x, None, mode), cin, c, h, w)
            x = self._conv(f"conv{i}_2", self._src(a, None, L.PRO_LEAKY), c, c, h, w)      # raw; consumers apply LeakyReLU
            cin, mode = c, L.PRO_LEAKY
            if i < 5:
                feats.append((x, h, w))
                ph, pw = (h + 1) // 2, (w + 1) // 2
                p = self._f(B, ph, pw, c)                                                 # max commutes with LeakyReLU
                self._add("nd_maxpool2x2_nhwc_f32", x.data_ptr(), p.data_ptr(), B, h, w, c, self.stream)
                x, h, w = p, ph, pw
        for i, c in zip(range(6, 10), reversed(LSID_STAGES[:-1])):
            skip, sh, sw = feats.pop()
            up = self._f(B, sh, sw, c)                      # ConvTranspose2d(2, s=2) + crop to the skip's size (:135)
            d = L.Pointwise()
            d.src, d.weight, d.out = self._src(x, None, L.PRO_LEAKY), self.w[f"up{i}.weight"].data_ptr(), up.data_ptr()
            d.B, d.HW, d.W, d.cin, d.cout, d.ldo = B, h * w, w, cin, 4 * c, c
            d.shuffle_c, d.shuffle_h, d.shuffle_w = c, sh, sw
            self._add("nd_pointwise_gemm_nhwc_f32", C.byref(d), self.stream)
            self.keep.append(d)
            h, w = sh, sw
            a = self._conv(f"conv{i}_1", self._src(up, skip, L.PRO_LEAKY_SECOND), 2 * c, c, h, w)
            x = self._conv(f"conv{i}_2", self._src(a, None, L.PRO_LEAKY), c, c, h, w)
            cin = c
        y = self._f(B, H, W, 4)
        d = L.Pointwise()
        d.src, d.weight, d.bias, d.out = self._src(x, None, L.PRO_LEAKY), self.w["conv10.weight"].data_ptr(), self.w["conv10.bias"].data_ptr(), y.data_ptr()
        d.B, d.HW, d.W, d.cin, d.cout, d.ldo = B, H * W, W, cin, 4, 4
        self._add("nd_pointwise_gemm_nhwc_f32", C.byref(d), self.stream)
        self.keep.append(d)
        self._add("nd_nhwc_to_nchw_f32", y.data_ptr(), self.out_nchw.data_ptr(), B, 4, H, W, self.stream)

    def run(self, x: torch.Tensor) -> torch.Tensor:
        with torch.cuda.device(self.dev):
            self.x_nchw.copy_(x.to(self.dev, torch.float32))
            torch.cuda.synchronize(self.dev)
            for fn, args, name in self.ops:
                r = fn(*args)
                if r != 0:
                    L.check(r, name)
            L.call("nd_stream_sync", self.stream)
            return self.out_nchw.clone()
