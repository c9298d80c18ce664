"""Host-side sequencing of the HIP kernels: NoiseDiffNet.forward and the sampling loop.

``Engine`` owns the packed weight arena (one flat fp32 tensor -- the only thing that has to be
broadcast to other ranks), ``Plan`` owns the workspace and the recorded launch list for one
(batch, height, width).  A plan is a list of ``(c_function, args)``; running it is a loop of
ctypes calls on the library's own HIP stream, and the per-step part is captured once into a
hipGraph and replayed for every diffusion step (the timestep and the sampler coefficients are
read by the kernels from device memory, so the captured graph never changes).

What is computed where (reference: models/archs/Diffusion_arch.py:577-646):
  * once per condition (``set_condition``): pos_emb (:584-585), the two ResnetBlock2 scale/shift
    maps (:188-190), iso embedding (:591) and every AttnBlock's per-sample vector
    to_out(to_v(iso)) -- CrossAttention over a 1-token context is exactly that vector
    (softmax over one key == 1), so to_q / to_k / norm1 never run;
  * once per step (``step_ops``): time MLP + all 20 ResnetBlock.mlp projections as one tall
    GEMV, then the U-Net body;
  * once per step, sampler: nd_sampler_step_* in place on the NHWC state.
PyTorch is used for allocation and host<->device copies only.
"""
from __future__ import annotations

import ctypes as C
import logging
import math
import os
import threading
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from .spec import (normalize_stage_attn, stage_attention_param_spec, ATTN_DIM_HEAD, ATTN_HEADS, ISO_DIM, ISO_TABLE_ROWS, POS_DIM, POS_GROUPS, RESNET_GROUPS, SHOT_GROUPS,
                   arch_param_spec, arch_traits, attention_param_spec, stage_dims)

GN_EPS = 1e-5
_log = logging.getLogger("noisediff_amd")
WINOGRAD = os.environ.get("ND_WINOGRAD", "1") != "0"     # tuning / A-B knob: 0 = direct conv3x3 kernel everywhere
PREACT = True            # wide block2 inputs get a separate GroupNorm + SiLU pass instead of the conv prologue (settled r1e; was ND_PREACT)
PREACT_MIN = 256         # ... from this width on (r1e: 128 / 256 / 512 measured the same within 0.1 %; was ND_PREACT_MIN)
CHAIN = True             # fused per-pixel Linear chains where pwchain.hip has the widths (was ND_CHAIN)
_CHAIN_FIRST = (".ff.net.0.0.weight", ".fc1.weight")     # Linear layers that can open / continue a fused chain (pwchain.hip)
_CHAIN_LATER = (".ff.net.2.weight", ".proj_out.weight", ".fc2.weight")
WINO2 = os.environ.get("ND_WINO2", "1") != "0"           # A-B knob: 0 = the two-waves-per-SIMD Winograd kernel (conv3x3_wino.hip)
MAP_BLOCKED = True       # ResnetBlock2 scale / shift maps in the 16-channel-blocked layout the F(4x4) kernel reads (was ND_MAP_BLOCKED)
# r6: ... or not stored at all -- the F(4x4) kernel forms them per chunk from silu(pos_emb) (8 channels) on the matrix pipe (ND_PRO_AFFINE_GENMAP_SILU): 2 x 33 MB per
# launch instead of 2 x 268, -50 us per launch, no map tensors (1 GB at 16 patches of 256 x 256).  ND_GENMAP=0: A/B knob (tools/ only): the stored maps.
GENMAP = os.environ.get("ND_GENMAP", "1") != "0"
COND_STEP = True         # time embedding / time_mlp / projections in one launch where it fits the LDS (was ND_COND_STEP)
WINO4 = os.environ.get("ND_WINO4", "1") != "0"           # A-B knob: 0 = never the F(4x4,3x3) kernel (conv3x3_wino4.hip)
# Low-latency mode for SMALL batches (opt-in: ND_SPLIT_K=1 or engine.SPLIT_K = True before the plans are recorded): plain-source F(4x4) layers whose
# (sample, region, cout tile) items fill a fraction of the chip run on the split-K form (nd_conv3x3_wino4_splitk_nhwc_f32: one sample at 256 x 256 has 16
# items for the 512 -> 512 layers at 32 x 32).  Off by default: the split count depends on the batch size, and with it the summation order over cin --
# a sample's bits would depend on the batch it is sharded into, which the default path rules out (see _wino4_takes).  16 patches per GPU never split.
SPLIT_K = os.environ.get("ND_SPLIT_K", "0") != "0"
EIGHT_TILES_RULE = True  # layers with <= 8 tiles of the 16 x 32 form per sample go to the 16 x 16-region form (profiles/r3_eight_tiles_rule_ab.txt; was ND_W4_EIGHT_TILES)
# r4: the F(4x4) kernel on 16 x 16-pixel regions with two co-resident workgroups per CU (nd_conv3x3_wino4_16_nhwc_f32).  "narrow" (default): where the
# 16 x 32 form has too few regions per sample or the image is narrower than 32 pixels (those layers ran on F(2x2) before); "all": wherever it takes the
# layer; "0": never.  Either way a function of the sample's geometry alone, and the same bits as the 16 x 32 form.
WINO4_16 = os.environ.get("ND_WINO4_16", "narrow")
WINO4_16_SPLIT = True    # geometry-only K ranges on the 16 x 16-region form (profiles/r4a_cfg2_cfg3_sampling_split_k_ab.txt; was ND_WINO4_16_SPLIT)
_ALIGN = 64   # floats; keeps every arena slice 256-byte aligned
# r6: wide 1x1 / Linear layers and the fused per-pixel chains on the bf16 matrix cores at full fp32 significand (three-term split of every operand, six
# products, fp32 accumulation: pointwise.hip / pwchain.hip, "SPLIT").  Same results within fp32 rounding (error against fp64 at or below the fp32 kernels',
# profiles/r6_split_gemm_accuracy.txt), 1.5-2x the speed: the fp32 matrix instruction issues on the VALU's own lanes.  ND_SPLIT_GEMM=0: A/B knob (tools/ only).
_SPLIT = os.environ.get("ND_SPLIT_GEMM", "1")                 # "1" both kinds, "chain" / "pw" one of them, "0" neither (A/B, tools/ only)
SPLIT_GEMM = _SPLIT != "0"
SPLIT_CHAIN, SPLIT_PW = _SPLIT in ("1", "chain"), _SPLIT in ("1", "pw")
TIME_TABLE = True        # the time embedding's head looked up per timestep (r3; was ND_TIME_TABLE)
PROJ_TABLE = True        # ... and the stacked ResnetBlock.mlp projection (r4e; was ND_PROJ_TABLE)
TIME_TABLE_ROWS = 1000                                      # timesteps the table covers (the reference's --timesteps; larger t: computed)


def _few_items(H: int, W: int, cout: int) -> bool:
    """At most eight (16 x 32-pixel region, 64-cout tile) items of the one-workgroup F(4x4) form per SAMPLE."""
    return ((H + 15) // 16) * ((W + 31) // 32) * ((cout + 63) // 64) <= 8


def _wino4_layer(cin: int, cout: int) -> bool:
    """Layers whose weights are also packed for the F(4x4,3x3) kernel: at least two 16-channel K chunks (conv3x3_wino4.hip)."""
    return cin > 16 and cin % 4 == 0 and cout % 4 == 0


_MAP_PRODUCERS = ("pos_block1.mlp.1.weight", "pos_block2.mlp.1.weight")


def _blocked_map_rows(C_: int) -> torch.Tensor:
    """Row order of a (2C, K) scale|shift producer for the blocked map layout: position 32 c + j holds scale channel 16 c + j
    (j < 16) or shift channel 16 c + j - 16 (include/noisediff_hip.h, nd_src.map_blocked)."""
    i = torch.arange(2 * C_)
    c, j = i // 32, i % 32
    return torch.where(j < 16, 16 * c + j, C_ + 16 * c + j - 16)


_CHAIN_SPLIT_LIMIT = (1 << 32) - (1 << 17)    # bytes of one tensor a launch of nd_pointwise_chain_split_nhwc_f32 addresses (32-bit buffer offsets)


def _split_layer(cin: int, cout: int) -> bool:
    """1x1 / Linear layers whose weights are also packed as three bf16 terms for nd_pointwise_gemm_split_nhwc_f32 (pointwise.hip: pw_split_takes)."""
    return SPLIT_PW and cin % 32 == 0 and cin >= 64 and cout % 128 == 0


def _classify(name: str, shape: Sequence[int]) -> str:
    """How a state-dict tensor is laid out in the arena."""
    if ".attn.to_q." in name or ".attn.to_k." in name or ".norm1." in name:
        return "skip"                       # dead with a 1-token context
    if len(shape) == 4:
        k = shape[-1]
        if name == "pos_enc.weights.weight" or name.endswith(".g"):
            return "raw"
        if k == 7:
            return "conv7"
        if k == 3:
            return "conv3"
        if name.startswith("downs.") and name.endswith((".3.1.weight", ".2.1.weight")):
            return "pw_unshuffle"           # Downsample = Rearrange + conv1x1 (stage index 3 with an AttnBlock, 2 without)
        return "pw"
    if len(shape) == 2 and (".ff.net." in name):
        return "pw"
    return "raw"


@dataclass
class Slot:
    offset: int
    numel: int
    kind: str
    shape: Tuple[int, ...]


class Engine:
    """Packed weights of one U-Net (NoiseDiffNet or one of the UNet_PosEmbV2* ablation nets) on one GPU."""

    def __init__(self, dim: int, device: torch.device, mid_attn: bool = False, inp_dim: int = 4, arch: str = "NoiseDiffNet", stage_attn=None):
        if device.type != "cuda":
            raise L.HipError("noisediff_amd runs on MI355X only: got device %r (there is no CPU path)" % (device,))
        self.lib = L.load()
        self.dim, self.device, self.mid_attn, self.inp_dim = dim, device, mid_attn, inp_dim
        self.arch, self.traits = arch, arch_traits(arch)
        self.spec = list(arch_param_spec(arch, dim, inp_dim))
        if mid_attn:
            self.spec += attention_param_spec("mid_attn", 8 * dim)
        self.stage_attn = normalize_stage_attn(stage_attn)         # per-stage LinearAttention / Attention (upstream's wiring; spec.py)
        if self.stage_attn:
            self.spec += stage_attention_param_spec(dim, self.stage_attn)
        self.resnet_names = [p.name[:-len(".mlp.1.weight")] for p in self.spec
                             if p.name.endswith(".mlp.1.weight") and len(p.shape) == 2]
        self.tproj_off: Dict[str, int] = {}
        self._layout()
        with torch.inference_mode(False):      # buffers outlive any inference_mode() block of the caller
            self.arena = torch.empty(self.arena_floats, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            s = C.c_void_p()
            L.call("nd_stream_create", C.byref(s))
        self.stream = s
        self.plans: Dict[Tuple[int, int, int], "Plan"] = {}
        self.loaded = False
        self.used: set = set()                     # arena slices the recorded plans read (Engine.p): what a broadcast has to carry
        self.valid: Optional[set] = None           # slices this rank holds after a partial broadcast (None: the whole arena)

    def __del__(self):                      # the engine owns its HIP stream (plans and loops only borrow it)
        s, self.stream = getattr(self, "stream", None), None
        for q in (s,):
            if q:
                try:
                    L.call("nd_stream_sync", q)
                    L.call("nd_stream_destroy", q)
                except Exception:           # interpreter shutdown: the library or the device may already be gone
                    pass

    # ------------------------------------------------------------------ arena layout
    def _layout(self) -> None:
        self.slots: Dict[str, Slot] = {}
        off = 0

        def add(name, numel, kind, shape):
            nonlocal off
            self.slots[name] = Slot(off, int(numel), kind, tuple(shape))
            off += (int(numel) + _ALIGN - 1) // _ALIGN * _ALIGN

        for p in self.spec:
            kind = _classify(p.name, p.shape)
            if kind == "skip":
                continue
            if kind == "conv3":
                n = self.lib.nd_pack_conv3x3_weight_floats(p.shape[1], p.shape[0])
                add(p.name + ".wino", self.lib.nd_pack_conv3x3_wino_weight_floats(p.shape[1], p.shape[0]), "derived", p.shape)
                if WINO4 and _wino4_layer(p.shape[1], p.shape[0]):
                    add(p.name + ".wino4", self.lib.nd_pack_conv3x3_wino4_weight_floats(p.shape[1], p.shape[0]), "derived", p.shape)
            elif kind in ("pw", "pw_unshuffle"):
                n = self.lib.nd_pack_pointwise_weight_floats(p.shape[1], p.shape[0])
                if p.name in _MAP_PRODUCERS and p.shape[0] % 32 == 0:
                    # ResnetBlock2.mlp[1] writes the per-pixel scale / shift map; a second copy with its output rows permuted writes
                    # the 16-channel-blocked layout the F(4x4,3x3) kernel reads one 128-byte line at a time (nd_src.map_blocked)
                    add(p.name + ".blk16", n, "derived", p.shape)
                    add(p.name[:-len("weight")] + "bias.blk16", p.shape[0], "derived", (p.shape[0],))
                    if GENMAP and p.shape[1] == POS_DIM == 8 and p.shape[0] % 32 == 0:
                        add(p.name + ".gen", p.shape[0] * p.shape[1], "derived", p.shape)      # the weight as it is, [2C][8]: the kernel's A operand
                if kind == "pw" and p.name.endswith(_CHAIN_FIRST + _CHAIN_LATER):
                    first = int(p.name.endswith(_CHAIN_FIRST))
                    add(p.name + ".chain", self.lib.nd_pack_chain_weight_floats(p.shape[1], p.shape[0], first), "derived", p.shape)
                    if SPLIT_CHAIN:
                        add(p.name + ".chain_s", self.lib.nd_pack_chain_weight_split_floats(p.shape[1], p.shape[0], first), "derived", p.shape)
                if kind == "pw" and _split_layer(p.shape[1], p.shape[0]):
                    add(p.name + ".split", self.lib.nd_pack_pointwise_weight_split_floats(p.shape[1], p.shape[0]), "derived", p.shape)
            elif kind == "conv7":
                n = 196 * p.shape[0]
                if SPLIT_PW and p.name == "init_conv.weight":      # the per-step stem on the split-product kernel (cond_init_conv runs once per condition)
                    add(p.name + ".split", self.lib.nd_pack_conv7x7_weight_split_floats(p.shape[0]), "derived", p.shape)
            else:
                n = math.prod(p.shape)
            add(p.name, n, kind, p.shape)
        # all ResnetBlock.mlp[1] Linears stacked into one (J, 4*dim) matrix
        j = 0
        for r in self.resnet_names:
            self.tproj_off[r] = j
            j += self.slots[r + ".mlp.1.weight"].shape[0]
        self.tproj_rows = j
        add("tproj.weight", j * 4 * self.dim, "derived", (j, 4 * self.dim))
        add("tproj.bias", j, "derived", (j,))
        add("time_freqs", self.dim // 2, "derived", (self.dim // 2,))
        # time_mlp's result for every timestep 0 .. TIME_TABLE_ROWS - 1 (st = SiLU(time_mlp(emb(t))), Diffusion_arch.py:100-107,502-507,149): the
        # step kernel looks the head up instead of running two dependent small Linears at the start of every diffusion step
        # (only where the table's builder runs: its head batch of 16 timesteps needs 16 * dim * 36 bytes of LDS and 4 * dim <= 2048 -- dim <= 280;
        #  wider nets keep nd_cond_step_f32 / the separate launches, as before r3)
        self.time_table = TIME_TABLE and 4 * self.dim <= 2048 and self.lib.nd_cond_step_lds_bytes(16, self.dim) <= 160 * 1024
        if self.time_table:
            add("time_table", TIME_TABLE_ROWS * 4 * self.dim, "derived", (TIME_TABLE_ROWS, 4 * self.dim))
            # ... and the stacked ResnetBlock.mlp projection of every tabulated timestep (41 MB at d = 64): a step's time conditioning is then a copy of B rows
            if PROJ_TABLE and self.tproj_rows % 4 == 0 and TIME_TABLE_ROWS * self.tproj_rows * 4 <= (192 << 20):
                add("tproj_table", TIME_TABLE_ROWS * self.tproj_rows, "derived", (TIME_TABLE_ROWS, self.tproj_rows))
        self.arena_floats = off

    def view(self, name: str) -> torch.Tensor:
        s = self.slots[name]
        return self.arena[s.offset:s.offset + s.numel]

    def p(self, name: str) -> int:
        s = self.slots[name]
        if self.valid is not None and name not in self.valid:
            raise L.HipError(f"arena slice {name!r} was not part of this rank's weight broadcast (it carried the slices of the plans "
                             "recorded before it): record the plan before Engine.broadcast(only_used=True), or broadcast the whole arena")
        self.used.add(name)
        return self.arena.data_ptr() + 4 * s.offset

    # ------------------------------------------------------------------ loading
    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """Pack a reference-layout state dict (any device) into the arena."""
        st = self.stream
        self.valid = None                          # every slice is (re)written below: an engine adopted from a partial broadcast may repack
        keep = []
        used = set(self.used)                      # packing addresses every slice; `used` records what the PLANS read
        with torch.cuda.device(self.device):
            for p in self.spec:
                if p.name not in self.slots:
                    continue
                t = sd[p.name].detach().to(device=self.device, dtype=torch.float32).contiguous()
                if tuple(t.shape) != tuple(p.shape):
                    raise ValueError(f"{p.name}: shape {tuple(t.shape)} != {tuple(p.shape)}")
                keep.append(t)
                kind = self.slots[p.name].kind
                if kind == "raw":
                    self.view(p.name).copy_(t.reshape(-1))
            torch.cuda.synchronize(self.device)     # raw copies ran on torch's stream
            for p, t in zip([q for q in self.spec if q.name in self.slots], keep):
                kind, dst = self.slots[p.name].kind, self.p(p.name)
                if kind == "conv3":
                    L.call("nd_pack_conv3x3_weight", t.data_ptr(), dst, p.shape[1], p.shape[0], st)
                    L.call("nd_pack_conv3x3_wino_weight", t.data_ptr(), self.p(p.name + ".wino"), p.shape[1], p.shape[0], st)
                    if p.name + ".wino4" in self.slots:
                        L.call("nd_pack_conv3x3_wino4_weight", t.data_ptr(), self.p(p.name + ".wino4"), p.shape[1], p.shape[0], st)
                elif kind == "pw":
                    L.call("nd_pack_pointwise_weight", t.data_ptr(), dst, p.shape[1], p.shape[0], 0, st)
                    if p.name + ".blk16" in self.slots:
                        perm = _blocked_map_rows(p.shape[0] // 2).to(self.device)
                        tp = t[perm].contiguous()
                        keep.append(tp)
                        L.call("nd_pack_pointwise_weight", tp.data_ptr(), self.p(p.name + ".blk16"), p.shape[1], p.shape[0], 0, st)
                        bname = p.name[:-len("weight")] + "bias"
                        self.view(bname + ".blk16").copy_(sd[bname].detach().to(device=self.device, dtype=torch.float32)[perm])
                    if p.name + ".gen" in self.slots:
                        self.view(p.name + ".gen").copy_(t.reshape(-1))
                    if p.name + ".chain" in self.slots:
                        L.call("nd_pack_chain_weight", t.data_ptr(), self.p(p.name + ".chain"), p.shape[1], p.shape[0],
                               int(p.name.endswith(_CHAIN_FIRST)), st)
                    if p.name + ".chain_s" in self.slots:
                        L.call("nd_pack_chain_weight_split", t.data_ptr(), self.p(p.name + ".chain_s"), p.shape[1], p.shape[0],
                               int(p.name.endswith(_CHAIN_FIRST)), st)
                    if p.name + ".split" in self.slots:
                        L.call("nd_pack_pointwise_weight_split", t.data_ptr(), self.p(p.name + ".split"), p.shape[1], p.shape[0], st)
                elif kind == "pw_unshuffle":
                    L.call("nd_pack_pointwise_weight", t.data_ptr(), dst, p.shape[1], p.shape[0], p.shape[1] // 4, st)
                elif kind == "conv7":
                    L.call("nd_pack_conv7x7_weight", t.data_ptr(), dst, p.shape[0], st)
                    if p.name + ".split" in self.slots:
                        L.call("nd_pack_conv7x7_weight_split", t.data_ptr(), self.p(p.name + ".split"), p.shape[0], st)
            L.call("nd_stream_sync", st)
            tw = torch.cat([self.view(r + ".mlp.1.weight") for r in self.resnet_names])
            tb = torch.cat([self.view(r + ".mlp.1.bias") for r in self.resnet_names])
            self.view("tproj.weight").copy_(tw)
            self.view("tproj.bias").copy_(tb)
            # SinusoidalPosEmb frequencies, computed exactly as Diffusion_arch.py:102-104 (fp32 on the host)
            half = self.dim // 2
            e = math.log(10000) / (half - 1)
            self.view("time_freqs").copy_(torch.exp(torch.arange(half) * -e).to(torch.float32))
            torch.cuda.synchronize(self.device)
            self._build_time_tables()
        self.loaded, self.valid, self.used = True, None, used

    _DERIVED_TABLES = ("time_table", "tproj_table")       # pure functions of time_freqs, time_mlp.* and tproj.*: never shipped, rebuilt where those arrive

    def _build_time_tables(self) -> None:
        """time_mlp's result and the stacked ResnetBlock.mlp projection for every tabulated timestep, from the arena's own time_freqs / time_mlp / tproj slices
        (the same kernel code the per-step launch runs: a lookup is the same bits).  Does not touch ``used``."""
        if not self.time_table:
            return
        st, used = self.stream, set(self.used)
        with torch.cuda.device(self.device):
            L.call("nd_cond_table_build_f32", self.p("time_freqs"), self.p("time_mlp.1.weight"), self.p("time_mlp.1.bias"), self.p("time_mlp.3.weight"),
                   self.p("time_mlp.3.bias"), self.p("time_table"), TIME_TABLE_ROWS, self.dim, st)
            L.call("nd_stream_sync", st)
            if "tproj_table" in self.slots:      # rows of the projection table = what the per-step launch computes for t = 0 .. rows-1
                ts = torch.arange(TIME_TABLE_ROWS, dtype=torch.int64, device=self.device)
                torch.cuda.synchronize(self.device)
                J = self.tproj_rows
                for t0 in range(0, TIME_TABLE_ROWS, 16):
                    nb = min(16, TIME_TABLE_ROWS - t0)
                    L.call("nd_cond_step_table_f32", ts.data_ptr() + 8 * t0, self.p("time_freqs"), self.p("time_mlp.1.weight"), self.p("time_mlp.1.bias"),
                           self.p("time_mlp.3.weight"), self.p("time_mlp.3.bias"), self.p("tproj.weight"), self.p("tproj.bias"),
                           self.p("tproj_table") + 4 * t0 * J, J, nb, self.dim, J, self.p("time_table"), TIME_TABLE_ROWS, st)
                L.call("nd_stream_sync", st)
        self.used = used

    def broadcast_state_dict(self, sd: Optional[Dict[str, torch.Tensor]], src: int = 0, group=None) -> int:
        """The ONE collective of the sampling path in its smallest form: the network's own fp32 weights (150 MB at d=64, 598 MB at
        d=128 -- SURVEY 8e) as one flat buffer, root -> all ranks; every rank then packs its arena itself (the packings are pure
        functions of the weights: F(4x4)-transformed 3x3 weights alone are 4x the raw ones, so shipping the packed arena costs 3-6x
        the bytes).  ``sd``: the state dict on rank ``src`` (ignored elsewhere).  Returns the bytes sent."""
        import torch.distributed as dist
        names = [p for p in self.spec if p.name in self.slots]
        total = sum(math.prod(p.shape) for p in names)
        with torch.cuda.device(self.device), torch.inference_mode(False):
            if dist.get_rank(group) == src:
                flat = torch.cat([sd[p.name].detach().reshape(-1).to(torch.float32) for p in names]).to(self.device)
                if flat.numel() != total:
                    raise ValueError("broadcast_state_dict: the state dict does not match the network's parameter spec")
            else:
                flat = torch.empty(total, dtype=torch.float32, device=self.device)
            torch.cuda.synchronize(self.device)
            dist.broadcast(flat, src=src, group=group)
            torch.cuda.synchronize(self.device)
            views, at = {}, 0
            for p in names:
                n = math.prod(p.shape)
                views[p.name] = flat[at:at + n].view(*p.shape)
                at += n
        self.load_state_dict(views)
        return total * 4

    def broadcast(self, src: int = 0, group=None, only_used: bool = False) -> int:
        """The ONE collective of the sampling path: packed weights root -> all ranks (RCCL over xGMI).  Returns the bytes sent.

        ``only_used``: carry only the arena slices the plans recorded so far read (``Engine.plan(..., allow_empty=True)`` records a
        plan on a rank that has no weights yet).  The arena holds up to three packings of every 3x3 weight (direct, F(2x2), F(4x4))
        plus chain / blocked copies; a given problem size reads one of them per layer -- at d=64, 256x256 that is 0.27 of the
        arena.  The slices travel as one gathered buffer (still a single broadcast); other slices are marked absent on the receivers."""
        import torch.distributed as dist
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)          # rank src: the packing kernels ran on the library's stream
            if not only_used:
                dist.broadcast(self.arena, src=src, group=group)
                nbytes = self.arena.numel() * 4
            else:
                # (the per-timestep tables are functions of slices that travel anyway -- 41 MB at d=64 that every receiver rebuilds instead)
                tables = [n for n in self._DERIVED_TABLES if n in self.used]
                if tables:
                    for dep in ("time_freqs", "time_mlp.1.weight", "time_mlp.1.bias", "time_mlp.3.weight", "time_mlp.3.bias", "tproj.weight", "tproj.bias"):
                        self.used.add(dep)
                names = sorted((n for n in self.used if n not in self._DERIVED_TABLES), key=lambda n: self.slots[n].offset)
                every = [None] * dist.get_world_size(group)
                dist.all_gather_object(every, names, group=group)      # control plane: every rank must have recorded the same plans
                if any(e != names for e in every):
                    raise L.HipError("Engine.broadcast(only_used=True): the ranks recorded different plans")
                ranges: List[List[int]] = []
                for n in names:
                    sl = self.slots[n]
                    lo, hi = sl.offset, sl.offset + (sl.numel + _ALIGN - 1) // _ALIGN * _ALIGN
                    if ranges and ranges[-1][1] == lo:
                        ranges[-1][1] = hi
                    else:
                        ranges.append([lo, hi])
                is_src = dist.get_rank(group) == src
                with torch.inference_mode(False):
                    stage = (torch.cat([self.arena[lo:hi] for lo, hi in ranges]) if is_src
                             else torch.empty(sum(hi - lo for lo, hi in ranges), dtype=torch.float32, device=self.device))
                dist.broadcast(stage, src=src, group=group)
                if not is_src:
                    at = 0
                    for lo, hi in ranges:
                        self.arena[lo:hi].copy_(stage[at:at + hi - lo])
                        at += hi - lo
                    self.valid = set(names) | set(tables)
                    if tables:
                        torch.cuda.synchronize(self.device)
                        self._build_time_tables()
                nbytes = stage.numel() * 4
            # the collective is asynchronous to the host and ordered only against torch's stream; the kernels that read the
            # arena run on the library's own non-blocking stream, so the arena must be complete before this returns
            torch.cuda.synchronize(self.device)
        self.loaded = True
        return nbytes

    # ------------------------------------------------------------------ plans
    def copy_from(self, other: "Engine") -> None:
        """Fill the arena from another device's engine of the same network (one device-to-device copy of the packed weights)."""
        if other.arena_floats != self.arena_floats or not other.loaded:
            raise L.HipError("Engine.copy_from: the source engine has no weights or another layout")
        other.sync()
        torch.cuda.synchronize(other.device)
        with torch.cuda.device(self.device):
            self.arena.copy_(other.arena)
            torch.cuda.synchronize(self.device)
        self.loaded, self.valid = True, other.valid

    def plan(self, B: int, H: int, W: int, debug: bool = False, tag: int = 0, allow_empty: bool = False) -> "Plan":
        """The plan (workspace + recorded launches) of one problem size; ``tag`` separates plans of equal size that must not share
        a workspace (several shards of one batch on the same device).  ``allow_empty``: record the plan before the weights arrive
        (recording needs the arena's layout only) -- see ``broadcast(only_used=True)``."""
        if not self.loaded and not allow_empty:
            raise L.HipError("Engine has no weights: call load_state_dict() or broadcast() first")
        key = (B, H, W, debug, tag)
        if key not in self.plans:
            self.plans[key] = Plan(self, B, H, W, debug)
        return self.plans[key]

    def sync(self) -> None:
        L.call("nd_stream_sync", self.stream)


Op = Tuple[Callable, tuple]


class Plan:
    """Workspace + recorded launch lists of one (B, H, W) problem."""

    def __init__(self, eng: Engine, B: int, H: int, W: int, debug: bool = False):
        # debug=True: no workspace reuse and named intermediates kept in self.taps (tests / diagnosis only)
        self.debug, self.taps = debug, {}
        self.lock = threading.Lock()           # callers that may share a plan across threads (NoiseDiffNet.forward under nn.DataParallel)
        if H % 8 or W % 8:
            raise AssertionError(f"your input dimensions {(H, W)} need to be divisible by 8, given the unet")  # :578
        self.e, self.B, self.H, self.W = eng, B, H, W
        self.dev = eng.device
        self._keep: List[object] = []          # ctypes structs / tensors referenced by recorded ops
        self._free: Dict[int, List[torch.Tensor]] = {}
        self.workspace_floats = 0
        d = eng.dim
        with torch.cuda.device(self.dev), torch.inference_mode(False):
            f = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.dev)
            # API-side buffers (NHWC state; NCHW mirrors are made on demand)
            self.x = f(B, H, W, eng.inp_dim)             # diffusion state x_t
            self.clean = f(B, H, W, eng.inp_dim)
            self.position = f(B, 2, H, W)                # NCHW as given
            self.iso_idx = torch.zeros(B, dtype=torch.int64, device=self.dev)
            self.time = torch.zeros(B, dtype=torch.int64, device=self.dev)
            self.model_out = f(B, H, W, eng.inp_dim)
            self.nchw_tmp = f(B, eng.inp_dim, H, W)
            self.pos_emb = f(B, H, W, POS_DIM)
            # the maps' consumers are pos_block{1,2}.block2.proj (d -> d, map prologue): formed inside conv3x3_wino4's 16 x 32-region form where that takes the layer
            # (a function of the sample's geometry alone), else stored -- in the blocked layout when the layer runs on conv3x3_wino4 at all
            self.posgen = (GENMAP and eng.traits.position and all(b + ".mlp.1.weight.gen" in eng.slots for b in ("pos_block1", "pos_block2")) and d % 16 == 0 and
                           self._wino4_kind("pos_block1.block2.proj", L.PRO_AFFINE_GENMAP_SILU, False, d, 0, d, 0, d, d, H, W,
                                            self._conv_rows(H, W, 0, d, 0, d, d, L.PRO_AFFINE_GENMAP_SILU)) == "wino4")
            self.posmap1 = self.posmap2 = None
            if self.posgen:
                self.pos_e = f(B, H, W, POS_DIM)             # silu(pos_emb): what ResnetBlock2.mlp = Sequential(SiLU, Conv2d) applies its 1x1 to (Diffusion_arch.py:177)
                self.pos_e_mad = torch.tensor([0.0, 1.0, 0.0], device=self.dev).repeat_interleave(POS_DIM).repeat(B).view(B, 3, POS_DIM).contiguous()
            else:
                self.posmap1 = f(B, H, W, 2 * d)
                self.posmap2 = f(B, H, W, 2 * d)
            self.posmap_blocked = (not self.posgen and MAP_BLOCKED and "pos_block1.mlp.1.weight.blk16" in eng.slots and
                                   self._wino4_takes("pos_block1.block2.proj", L.PRO_AFFINE_MAP_SILU, False, d, 0, d, 0, d, d, H, W))
            self.iso_emb = f(B, ISO_DIM)
            self.attn_names = [p.name[:-len(".attn.to_v.weight")] for p in eng.spec if p.name.endswith(".attn.to_v.weight")]
            self.cb = {n: f(B, eng.slots[n + ".proj_out.bias"].shape[0]) for n in self.attn_names}
            self.emb = f(B, d)
            self.t1 = f(B, 4 * d)
            self.st = f(B, 4 * d)
            self.tproj = f(B, eng.tproj_rows)
        self.cond_ops: List[Op] = []
        self.step_ops: List[Op] = []
        self._ops = self.cond_ops
        self._record_condition()
        self._ops = self.step_ops
        self._record_time()
        self._record_net()
        self.condition_set = False

    # ------------------------------------------------------------------ tiny allocator
    def _alloc(self, *shape) -> torch.Tensor:
        n = math.prod(shape)
        lst = self._free.get(n)
        if lst:
            return lst.pop().view(*shape)
        self.workspace_floats += n
        with torch.cuda.device(self.dev), torch.inference_mode(False):
            t = torch.empty(n, dtype=torch.float32, device=self.dev)
        self._keep.append(t)
        return t.view(*shape)

    def _tap(self, name: str, t: torch.Tensor) -> torch.Tensor:
        if self.debug:
            self.taps[name] = t
        return t

    def _release(self, *ts: torch.Tensor) -> None:
        if self.debug:
            return
        for t in ts:
            self._free.setdefault(t.numel(), []).append(t.reshape(-1))

    # ------------------------------------------------------------------ op recording
    def _add(self, name: str, *args, meta: Optional[dict] = None) -> None:
        fn = getattr(self.e.lib, name)
        self._keep.append(args)
        self._ops.append((fn, args, name, meta))

    def _src(self, t: torch.Tensor, t2: Optional[torch.Tensor] = None, mode=L.PRO_NONE, **kw) -> L.Src:
        s = L.Src()
        s.p0, s.c0, s.ld0 = t.data_ptr(), t.shape[-1], t.shape[-1]
        if t2 is not None:
            s.p1, s.c1, s.ld1 = t2.data_ptr(), t2.shape[-1], t2.shape[-1]
        s.mode = mode
        for k, v in kw.items():
            setattr(s, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
        return s

    def _conv_rows(self, H: int, W: int, up: int, ld0: int, ld1: int, cin: int, cout: int, mode: int, splits: int = 1) -> int:
        """Samples ONE conv3x3 launch covers.  The Winograd kernels address their sources through 32-bit buffer offsets (sources below 1 GiB and 2^24 pixels,
        outputs below 4 GiB, split-K partial sums below 2 GiB): a larger batch is cut into equal pieces of whole samples -- same kernel, same weights, same
        summation order per sample, so a sample's bits do not depend on the batch (the reference takes any --batch_size, test_diffusion.py:23-78)."""
        sh, sw, lim = H >> up, W >> up, (1 << 30) - (1 << 16)

        def fits(b: int) -> bool:
            px = b * sh * sw + sw + 2
            return (px * 4 * max(ld0, ld1) < lim and px < (1 << 24) and b * H * W < (1 << 24) and b * H * W * cout * 4 < (1 << 32) - (1 << 16)
                    and (mode != L.PRO_AFFINE_MAP_SILU or (b * H + 2) * W * 8 * cin < lim) and splits * b * H * W * cout * 4 < (1 << 31))
        if fits(self.B):
            return self.B
        lo, hi = 1, self.B                         # largest b that fits (fits is monotone), then equal pieces
        while lo < hi:
            mid = (lo + hi + 1) // 2
            lo, hi = (mid, hi) if fits(mid) else (lo, mid - 1)
        if not fits(lo):
            raise L.HipError(f"conv3x3 at {H} x {W} x {max(ld0, ld1)}: one sample exceeds the kernels' 1 GiB source limit")
        pieces = -(-self.B // lo)
        return -(-self.B // pieces)

    def conv3(self, name: str, src: L.Src, cin: int, cout: int, H: int, W: int, stats: bool):
        """nn.Conv2d(cin, cout, 3, padding=1); returns (out, stats, slot_count, slots)."""
        e = self.e
        out = self._alloc(self.B, H, W, cout)
        # images of at least one 16x16 tile go through the Winograd F(2x2,3x3) kernel (2.25x fewer MFMAs)
        wino = WINOGRAD and H >= 16 and W >= 16
        up = 1 if src.upsample else 0
        # rows per launch (the whole batch at the bench sizes).  Every kernel choice below looks at ONE launch's piece, and the pieces are equal for every
        # batch beyond the limit: the choice is a function of the sample's geometry alone
        rows = self._conv_rows(H, W, up, src.ld0, src.ld1, cin, cout, src.mode) if wino else self.B
        # both Winograd kernels share weights, statistics slots and descriptor; wino2 (one resident wave per SIMD,
        # all 16 position accumulators in registers) is the default, wino covers what it does not take
        wino2 = (wino and WINO2 and not (src.mode == L.PRO_AFFINE_MAP_SILU and src.upsample)
                 and (src.c1 == 0 or src.c0 % 32 == 0) and rows * H * W < (1 << 24)
                 and rows * H * W * 4 * max(src.ld0, src.ld1, 2 * cin if src.mode == L.PRO_AFFINE_MAP_SILU else 0) < (1 << 31))
        w4kind = self._wino4_kind(name, src.mode, bool(src.upsample), src.c0, src.c1, src.ld0, src.ld1, cin, cout, H, W, rows) if wino else ""
        wino4 = bool(w4kind)
        if src.mode == L.PRO_AFFINE_GENMAP_SILU and w4kind != "wino4":
            raise L.HipError(f"{name}: maps formed in the kernel need conv3x3_wino4's 16 x 32-region form (Plan.posgen decides by the same rule)")
        if wino and not wino4 and (name + ".weight.wino4") in e.slots and src.mode in (L.PRO_NONE, L.PRO_AFFINE_SILU, L.PRO_AFFINE_MAP_SILU) and not _few_items(H, W, cout):
            _log.warning("%s (%d -> %d at %d x %d, batch %d): not on the F(4x4) kernel", name, cin, cout, H, W, self.B)
        st = sc = None
        slots = 0
        if stats:     # per-(slot, channel) {sum, M2} partials for nd_groupnorm_finalize_f32: F(4x4) one slot per 16 x 16 tile, F(2x2) two
            slots = (e.lib.nd_conv3x3_wino4_stat_slots(H, W) if wino4 else e.lib.nd_conv3x3_wino_stat_slots(H, W) if wino
                     else e.lib.nd_conv3x3_stat_slots(H, W, cout, self.B))
            st = self._alloc(self.B, slots, cout, 2)
            sc = self._alloc(slots)
        if src.map_blocked and not wino4:
            raise L.HipError(f"{name}: the scale / shift map was produced in the blocked layout but the layer does not run on conv3x3_wino4")
        # ONE packing of the weight is read (and marked for the weight broadcast): F(4x4), F(2x2) or the direct form
        lowlat_split = SPLIT_K and w4kind == "wino4" and src.mode in (L.PRO_NONE, L.PRO_AFFINE_SILU) and int(e.lib.nd_conv3x3_wino4_splitk_plan(self.B, H, W, cin, cout)) > 1
        weight = e.p(name + (".weight.wino4" if wino4 else ".weight.wino" if wino else ".weight"))
        entry = (f"nd_conv3x3_{w4kind}_nhwc_f32" if wino4 else "nd_conv3x3_wino2_nhwc_f32" if wino2 else
                 "nd_conv3x3_wino_nhwc_f32" if wino else "nd_conv3x3_nhwc_f32")
        tiling = (9016 if w4kind == "wino4_16" else 9004) if wino4 else 9002 if wino2 else 9001 if wino else e.lib.nd_conv3x3_tiling_id(self.B, H, W, cout)
        splits = int(e.lib.nd_conv3x3_wino4_splitk_plan(self.B, H, W, cin, cout)) if lowlat_split else 1
        if w4kind == "wino4_16" and WINO4_16_SPLIT:      # few items per sample: K ranges by the SAMPLE's geometry (the batch never enters: a sample's bits stay batch-invariant)
            splits = int(e.lib.nd_conv3x3_wino4_16_splitk_plan(H, W, cin, cout))
        if splits > 1:                                   # the partial sums of a launch stay below 2 GiB: smaller pieces, never another kernel
            rows = self._conv_rows(H, W, up, src.ld0, src.ld1, cin, cout, src.mode, splits)
        ws = self._alloc(splits * rows * H * W * cout) if splits > 1 else None
        sh, sw, ctot = H >> up, W >> up, src.c0 + src.c1
        for b0 in range(0, self.B, rows):                # one launch per piece of `rows` samples (one piece at the bench sizes)
            nb = min(rows, self.B - b0)
            d = L.Conv3x3()
            s = L.Src.from_buffer_copy(src)
            if b0:                                       # the piece's rows of every per-sample operand
                s.p0 = src.p0 + 4 * b0 * sh * sw * src.ld0
                if src.p1:
                    s.p1 = src.p1 + 4 * b0 * sh * sw * src.ld1
                if src.mad:
                    s.mad = src.mad + 4 * b0 * 3 * ctot
                if src.map:
                    s.map = src.map + 4 * b0 * H * W * (POS_DIM if src.mode == L.PRO_AFFINE_GENMAP_SILU else 2 * ctot)
            d.src, d.weight, d.bias, d.out = s, weight, e.p(name + ".bias"), out.data_ptr() + 4 * b0 * H * W * cout
            d.B, d.H, d.W, d.cin, d.cout, d.ldo = nb, H, W, cin, cout, cout
            if stats:
                d.stats, d.slot_count = st.data_ptr() + 4 * b0 * slots * cout * 2, sc.data_ptr()
            meta = {"layer": name, "B": nb, "H": H, "W": W, "cin": cin, "cout": cout, "mode": int(src.mode), "tiling": tiling}
            if splits > 1:
                meta["splits"] = splits
                entry_s = "nd_conv3x3_wino4_16_splitk_nhwc_f32" if w4kind == "wino4_16" else "nd_conv3x3_wino4_splitk_nhwc_f32"
                self._add(entry_s, C.byref(d), ws.data_ptr(), splits, e.stream, meta=meta)
            else:
                self._add(entry, C.byref(d), e.stream, meta=meta)
            self._keep.append(d)
        if ws is not None:
            self._release(ws)
        return out, st, sc, slots

    def _wino4_kind(self, name: str, mode: int, upsample: bool, c0: int, c1: int, ld0: int, ld1: int, cin: int, cout: int, H: int, W: int, rows: Optional[int] = None) -> str:
        """"wino4" (16 x 32-pixel regions, one workgroup per CU), "wino4_16" (16 x 16-pixel regions, two per CU) or "" (another kernel).
        F(4x4,3x3) (1.78x fewer MFMAs than F(2x2,3x3); 16-channel K chunks) wherever a form of its kernel takes the
        layer: plain / GroupNorm-affine (+ per-pixel map) + SiLU inputs, concat on a chunk boundary, images that fill its regions,
        sources below 1 GiB and 2^24 pixels (the host checks of nd_conv3x3_wino4_nhwc_f32)."""
        up = 1 if upsample else 0
        rows = self.B if rows is None else rows                          # samples of one launch (conv3 cuts a batch beyond the kernels' limits into pieces)
        src_px = rows * (H >> up) * (W >> up) + (W >> up) + 2            # the kernel's buffer resource starts one row + one pixel in front of the tensor
        src_bytes = src_px * 4 * max(ld0, ld1)
        common = (WINOGRAD and WINO4 and H >= 16 and W >= 16 and cout <= 2048 and (name + ".weight.wino4") in self.e.slots
                  and W <= 2048 and (c1 == 0 or (c0 % 16 == 0 and not up)) and src_bytes < (1 << 30) - (1 << 16) and src_px < (1 << 24)
                  and (not up or (H % 2 == 0 and W % 2 == 0)))
        if not common:
            return ""
        # the 16 x 16-region form: plain and affine + SiLU sources; regions of any image are at least half used from W = 16 on (W % 16 == 0 or W >= 48)
        takes16 = WINO4_16 != "0" and mode in (L.PRO_NONE, L.PRO_AFFINE_SILU) and (W % 16 == 0 or W >= 48)
        few = _few_items(H, W, cout)                                     # <= 8 items of the 16 x 32 form per SAMPLE
        if takes16 and (WINO4_16 == "all" or W < 32 or (few and EIGHT_TILES_RULE and not SPLIT_K)):
            return "wino4_16"
        return "wino4" if self._wino4_takes(name, mode, upsample, c0, c1, ld0, ld1, cin, cout, H, W, rows) else ""

    def _wino4_takes(self, name: str, mode: int, upsample: bool, c0: int, c1: int, ld0: int, ld1: int, cin: int, cout: int, H: int, W: int, rows: Optional[int] = None) -> bool:
        """The 16 x 32-region form."""
        up = 1 if upsample else 0
        # Layers with at most eight F(4x4) workgroup tiles (16 x 32 pixels x 64 couts) per SAMPLE -- 256 -> 256 at 32 x 32 -- fill half of an
        # MI355X at the usual 16 patches per GPU; F(2x2)'s 16 x 16-pixel tiles fill it (84 vs 114 us per layer).  The rule looks at the
        # sample's geometry only: the kernel choice -- and with it the bits of a sample -- must not depend on the batch it is sharded into.
        # (The opt-in low-latency mode, SPLIT_K, gives that up anyway and cuts these layers along cin instead.)
        if (WINO2 and not SPLIT_K and EIGHT_TILES_RULE and ((H + 15) // 16) * ((W + 31) // 32) * ((cout + 63) // 64) <= 8 and (c1 == 0 or c0 % 32 == 0)
                and not (mode == L.PRO_AFFINE_MAP_SILU and up)):
            return False
        rows = self.B if rows is None else rows
        src_px = rows * (H >> up) * (W >> up) + (W >> up) + 2            # the kernel's buffer resource starts one row + one pixel in front of the tensor
        src_bytes = src_px * 4 * max(ld0, ld1)
        return (WINOGRAD and WINO4 and H >= 16 and W >= 16 and cout <= 2048 and (name + ".weight.wino4") in self.e.slots
                and mode in (L.PRO_NONE, L.PRO_AFFINE_SILU, L.PRO_AFFINE_MAP_SILU, L.PRO_AFFINE_GENMAP_SILU)
                and not (mode == L.PRO_AFFINE_MAP_SILU and (up or (rows * H + 2) * W * 8 * cin >= (1 << 30) - (1 << 16)))
                and not (mode == L.PRO_AFFINE_GENMAP_SILU and (up or c1 or cin % 16))
                and W >= 32 and (W % 32 == 0 or W >= 96) and W <= 2048 and (c1 == 0 or (c0 % 16 == 0 and not up))
                and src_bytes < (1 << 30) - (1 << 16) and src_px < (1 << 24)
                and (not up or (H % 2 == 0 and W % 2 == 0)))

    def pw(self, name: str, src: L.Src, cin: int, cout: int, HW: int, W: int, act=L.ACT_NONE, bias=True,
           res0=None, res1=None, vec=None, gn_t=None, gn_mad=None, out=None, variant: str = "") -> torch.Tensor:
        """1x1 conv / token Linear with fused prologue/epilogue (`variant`: suffix of a permuted copy of weight and bias)."""
        e = self.e
        if out is None:
            out = self._alloc(self.B, HW, cout)
        d = L.Pointwise()
        d.src, d.out = src, out.data_ptr()
        d.bias = e.p(name + ".bias" + variant) if bias else None
        d.B, d.HW, d.W, d.cin, d.cout, d.ldo, d.act = self.B, HW, W, cin, cout, cout, act
        # wide layers: the split-product kernel on the bf16 matrix cores (a function of the layer's shape alone, never of the batch)
        split = not variant and (name + ".weight.split") in e.slots and bool(e.lib.nd_pointwise_gemm_split_takes(C.byref(d)))
        d.weight = e.p(name + (".weight.split" if split else ".weight" + variant))
        if res0 is not None:
            d.res0, d.ldr0 = res0.data_ptr(), res0.shape[-1]
        if res1 is not None:
            d.res1, d.ldr1 = res1.data_ptr(), res1.shape[-1]
        if vec is not None:
            d.vec = vec.data_ptr()
        if gn_t is not None:
            d.gn_t, d.ldt, d.gn_mad = gn_t.data_ptr(), gn_t.shape[-1], gn_mad.data_ptr()
        self._add("nd_pointwise_gemm_split_nhwc_f32" if split else "nd_pointwise_gemm_nhwc_f32", C.byref(d), e.stream,
                  meta={"layer": name, "B": self.B, "HW": HW, "cin": cin, "cout": cout, "split": int(split)})
        self._keep.append(d)
        return out

    def gn_finalize(self, st, sc, slots: int, norm: str, C_: int, groups: int, ss_off: Optional[int]) -> torch.Tensor:
        e = self.e
        mad = self._alloc(self.B, 3, C_)
        ss = (self.tproj.data_ptr() + 4 * ss_off) if ss_off is not None else None
        self._add("nd_groupnorm_finalize_f32", st.data_ptr(), sc.data_ptr(), slots, e.p(norm + ".weight"), e.p(norm + ".bias"),
                  ss, e.tproj_rows, mad.data_ptr(), self.B, C_, groups, GN_EPS, e.stream)
        return mad

    def linear_rows(self, x: torch.Tensor, w: int, b: Optional[int], out: torch.Tensor, K: int, N: int, act_in=0, act_out=0):
        self._add("nd_linear_rows_f32", x.data_ptr(), x.shape[-1], w, b, out.data_ptr(), out.shape[-1], self.B, K, N,
                  act_in, act_out, self.e.stream)

    # ------------------------------------------------------------------ composite layers
    def resnet(self, name: str, x: torch.Tensor, skip: Optional[torch.Tensor], cout: int, H: int, W: int, groups: int,
               posmap: Optional[torch.Tensor] = None, extra_res: Optional[torch.Tensor] = None) -> torch.Tensor:
        """ResnetBlock / ResnetBlock2 (Diffusion_arch.py:146-196) as 2 convs + 2 tiny finalizes + 1 tail."""
        cin = x.shape[-1] + (skip.shape[-1] if skip is not None else 0)
        HW = H * W
        c1, st1, sc1, n1 = self.conv3(name + ".block1.proj", self._src(x, skip), cin, cout, H, W, stats=True)
        ss_off = None if posmap is not None else self.e.tproj_off.get(name)     # blocks built with time_emb_dim=None have no mlp
        mad1 = self.gn_finalize(st1, sc1, n1, name + ".block1.norm", cout, groups, ss_off)
        a1 = None
        if PREACT and posmap is None and cout >= PREACT_MIN and H >= 16 and W >= 16:      # (same-box A/B at r1e: 128 -> 23.15, 256 -> 23.13, 512 -> 23.13, never -> 23.22 ms per step)
            # every 64-channel output slice of block2 would re-apply GroupNorm+SiLU to the same input tile (cout/64 times);
            # on these small, wide tensors one elementwise pass (tens of MB) is cheaper than that VALU work next to the MFMAs
            a1 = self._alloc(self.B, HW, cout)
            self._add("nd_affine_silu_add_f32", c1.data_ptr(), cout, mad1.data_ptr(), None, cout, None, cout, a1.data_ptr(), cout,
                      self.B, HW, cout, self.e.stream,
                      meta={"layer": name + ".block1.act", "stream_bytes": 4.0 * self.B * HW * cout * 2})
            src2 = self._src(a1)
        else:
            if posmap is not None and self.posgen:           # scale | shift formed in the kernel from silu(pos_emb) and this block's mlp[1]
                src2 = self._src(c1, None, L.PRO_AFFINE_GENMAP_SILU, mad=mad1, map=self.pos_e, gamma=self.e.p(name + ".mlp.1.weight.gen"),
                                 beta=self.e.p(name + ".mlp.1.bias"))
            else:
                mode = L.PRO_AFFINE_MAP_SILU if posmap is not None else L.PRO_AFFINE_SILU
                src2 = self._src(c1, None, mode, mad=mad1, **({"map": posmap, "map_blocked": int(self.posmap_blocked)} if posmap is not None else {}))
        c2, st2, sc2, n2 = self.conv3(name + ".block2.proj", src2, cout, cout, H, W, stats=True)
        if a1 is not None:
            self._release(a1)
        mad2 = self.gn_finalize(st2, sc2, n2, name + ".block2.norm", cout, groups, None)
        if cin != cout:       # h + res_conv(x): the 1x1 GEMM adds silu(GN(c2)) in its epilogue   :170
            assert extra_res is None
            out = self.pw(name + ".res_conv", self._src(x, skip), cin, cout, HW, W, gn_t=c2, gn_mad=mad2)
        else:
            assert skip is None
            out = self._alloc(self.B, HW, cout)
            self._add("nd_affine_silu_add_f32", c2.data_ptr(), cout, mad2.data_ptr(), x.data_ptr(), cout,
                      extra_res.data_ptr() if extra_res is not None else None, cout, out.data_ptr(), cout,
                      self.B, HW, cout, self.e.stream,
                      meta={"layer": name + ".tail", "stream_bytes": 4.0 * self.B * HW * cout * (4 if extra_res is not None else 3)})
        self._release(c1, st1, sc1, mad1, c2, st2, sc2, mad2)
        return out.view(self.B, H, W, cout)

    def chain(self, name: str, src: L.Src, stages, HW: int) -> torch.Tensor:
        """Fused per-pixel Linear chain; stages = [(layer, cin, cout, act, res)]."""
        e = self.e
        out = self._alloc(self.B, HW, stages[-1][2])
        # the split-product form (a function of the widths alone); two sources only where the first stage is one K step (pwchain.hip)
        split = all((layer + ".weight.chain_s") in e.slots for layer, *_ in stages) and (src.c1 == 0 or stages[0][1] <= 16)
        ldo = stages[-1][2]
        # the split kernel addresses its tensors through 32-bit buffer offsets (below 4 GiB): a larger batch runs as equal pieces of whole samples -- same kernel,
        # same arithmetic per pixel, so a sample's bits do not depend on the batch (as Plan._conv_rows does for the convolutions)
        rows = self.B
        if split:
            per = 4 * HW * max(src.ld0, src.ld1, ldo)
            rows = max(1, min(self.B, _CHAIN_SPLIT_LIMIT // per))
            rows = -(-self.B // -(-self.B // rows))
        for b0 in range(0, self.B, rows):
            nb = min(rows, self.B - b0)
            d = L.Chain()
            sp = L.Src.from_buffer_copy(src)
            if b0:
                sp.p0 = src.p0 + 4 * b0 * HW * src.ld0
                if src.p1:
                    sp.p1 = src.p1 + 4 * b0 * HW * src.ld1
                if src.vec:
                    sp.vec = src.vec + 4 * b0 * (src.c0 + src.c1)
            d.src, d.out, d.n_stages, d.B, d.HW, d.ldo = sp, out.data_ptr() + 4 * b0 * HW * ldo, len(stages), nb, HW, ldo
            for i, (layer, cin, cout, act, res) in enumerate(stages):
                d.st[i].weight, d.st[i].bias = e.p(layer + (".weight.chain_s" if split else ".weight.chain")), e.p(layer + ".bias")
                d.st[i].cin, d.st[i].cout, d.st[i].act, d.st[i].res = cin, cout, act, res
            self._add("nd_pointwise_chain_split_nhwc_f32" if split else "nd_pointwise_chain_nhwc_f32", C.byref(d), e.stream,
                      meta={"layer": name, "B": nb, "HW": HW, "cin": stages[0][1], "cout": stages[-1][2], "split": int(split),
                            "flop_per_px": 2.0 * sum(c_in * c_out for _, c_in, c_out, _, _ in stages)})
            self._keep.append(d)
        return out

    def _chain_ok(self, HW: int, widths) -> bool:
        w = list(widths) + [0] * (4 - len(widths))
        return CHAIN and HW % 32 == 0 and bool(self.e.lib.nd_pointwise_chain_supported(*w))

    def attn_block(self, name: str, x: torch.Tensor, H: int, W: int) -> torch.Tensor:
        """AttnBlock (Diffusion_arch.py:434-443) with the 1-token CrossAttention folded into cb."""
        Cc, HW = x.shape[-1], H * W
        cb = self.cb[name]
        if self._chain_ok(HW, (Cc, 2 * Cc, Cc, Cc)):
            ln = self._src(x, None, L.PRO_LAYERNORM, vec=cb, gamma=self.e.p(name + ".norm2.weight"), beta=self.e.p(name + ".norm2.bias"))
            y = self.chain(name + ".ff+proj_out", ln,
                           [(name + ".ff.net.0.0", Cc, 2 * Cc, L.ACT_GELU, L.CHAIN_RES_NONE),
                            (name + ".ff.net.2", 2 * Cc, Cc, L.ACT_NONE, L.CHAIN_RES_INPUT),
                            (name + ".proj_out", Cc, Cc, L.ACT_NONE, L.CHAIN_RES_INPUT_RAW)], HW)
            return y.view(self.B, H, W, Cc)
        kw = {}
        rs = None
        if Cc > 64:      # wide rows: per-pixel {mean, rstd} from a streaming pre-pass (narrow rows: derived inside the GEMM)
            rs = self._alloc(self.B, HW, 2)
            self._add("nd_layernorm_stats_f32", x.data_ptr(), Cc, cb.data_ptr(), rs.data_ptr(), self.B, HW, Cc, 1e-5, self.e.stream)
            kw["rowstats"] = rs
        ln = self._src(x, None, L.PRO_LAYERNORM, vec=cb, gamma=self.e.p(name + ".norm2.weight"), beta=self.e.p(name + ".norm2.bias"), **kw)
        h1 = self.pw(name + ".ff.net.0.0", ln, Cc, 2 * Cc, HW, W, act=L.ACT_GELU)
        x2 = self.pw(name + ".ff.net.2", self._src(h1), 2 * Cc, Cc, HW, W, res0=x, vec=cb)
        y = self.pw(name + ".proj_out", self._src(x2), Cc, Cc, HW, W, res0=x)
        self._release(h1, x2)
        if rs is not None:
            self._release(rs)
        return y.view(self.B, H, W, Cc)

    def mlp(self, name: str, src: L.Src, cin: int, hid: int, cout: int, H: int, W: int, res0=None) -> torch.Tensor:
        if res0 is None and src.mode == L.PRO_NONE and self._chain_ok(H * W, (cin, hid, cout)):
            o = self.chain(name, src, [(name + ".fc1", cin, hid, L.ACT_GELU, L.CHAIN_RES_NONE),
                                       (name + ".fc2", hid, cout, L.ACT_NONE, L.CHAIN_RES_NONE)], H * W)
            return o.view(self.B, H, W, cout)
        h = self.pw(name + ".fc1", src, cin, hid, H * W, W, act=L.ACT_GELU)
        o = self.pw(name + ".fc2", self._src(h), hid, cout, H * W, W, res0=res0)
        self._release(h)
        return o.view(self.B, H, W, cout)

    # ------------------------------------------------------------------ recorded programs
    def _record_condition(self) -> None:
        e, B, H, W, st = self.e, self.B, self.H, self.W, self.e.stream
        tr = e.traits
        if tr.position:
            pe = self._alloc(B, H, W, 3 * POS_DIM)
            self._add("nd_pos_enc_f32", self.position.data_ptr(), e.p("pos_enc.weights.weight"), e.p("pos_enc.weights.bias"),
                      pe.data_ptr(), B, H, W, POS_DIM, st)
            h = self.pw("pos_mlp.fc1", self._src(pe), 3 * POS_DIM, 2 * POS_DIM, H * W, W, act=L.ACT_GELU)
            self.pw("pos_mlp.fc2", self._src(h), 2 * POS_DIM, POS_DIM, H * W, W, out=self.pos_emb)
            if self.posgen:                                  # the maps are formed by their consumers: only the activation is shared work
                self._add("nd_affine_silu_add_f32", self.pos_emb.data_ptr(), POS_DIM, self.pos_e_mad.data_ptr(), None, POS_DIM, None, POS_DIM,
                          self.pos_e.data_ptr(), POS_DIM, B, H * W, POS_DIM, st, meta={"layer": "pos_emb.silu", "stream_bytes": 8.0 * B * H * W * POS_DIM})
            for blk, dst in (() if self.posgen else (("pos_block1", self.posmap1), ("pos_block2", self.posmap2))):
                self.pw(blk + ".mlp.1", self._src(self.pos_emb, None, L.PRO_SILU), POS_DIM, 2 * e.dim, H * W, W, out=dst,
                        variant=".blk16" if self.posmap_blocked else "")
            self._release(pe, h)
        if tr.iso_attn:
            self._add("nd_embedding_rows_f32", self.iso_idx.data_ptr(), e.p("iso_embed.weight"), self.iso_emb.data_ptr(), B,
                      ISO_TABLE_ROWS, ISO_DIM, st)
            inner = ATTN_HEADS * ATTN_DIM_HEAD
            v = self._alloc(B, inner)
            for n in self.attn_names:
                Cc = self.cb[n].shape[-1]
                self.linear_rows(self.iso_emb, e.p(n + ".attn.to_v.weight"), None, v, ISO_DIM, inner)
                self.linear_rows(v, e.p(n + ".attn.to_out.0.weight"), e.p(n + ".attn.to_out.0.bias"), self.cb[n], inner, Cc)
            self._release(v)
        self.clean_emb = None
        if tr.cond_branch:
            # clean-image encoding of the UNet_PosEmbV2* nets (others_arch.py:491-492): step-invariant, so it runs here
            ce = self._alloc(B, H, W, e.dim)
            self._add("nd_conv7x7_c4_f32", self.clean.data_ptr(), e.p("cond_init_conv.weight"), e.p("cond_init_conv.bias"),
                      ce.data_ptr(), e.dim, B, H, W, e.dim, st)
            self.clean_emb = self._tap("clean_emb", self.resnet("cond_res_block1", ce, None, e.dim, H, W, RESNET_GROUPS))
            self._release(ce)                  # clean_emb itself is never released: every step reads it

    def _record_time(self) -> None:
        e, d = self.e, self.e.dim
        if COND_STEP and 4 * d <= 2048 and e.lib.nd_cond_step_lds_bytes(self.B, d) <= 160 * 1024:
            # one launch: time embedding, time_mlp, SiLU and every ResnetBlock.mlp projection (tproj)
            args = (self.time.data_ptr(), e.p("time_freqs"), e.p("time_mlp.1.weight"), e.p("time_mlp.1.bias"),
                    e.p("time_mlp.3.weight"), e.p("time_mlp.3.bias"), e.p("tproj.weight"), e.p("tproj.bias"), self.tproj.data_ptr(),
                    e.tproj_rows, self.B, d, e.tproj_rows)
            if e.time_table and "tproj_table" in e.slots and e.tproj_rows % 4 == 0:
                self._add("nd_cond_step_ptable_f32", *args, e.p("time_table"), TIME_TABLE_ROWS, e.p("tproj_table"), e.stream)
            elif e.time_table:
                self._add("nd_cond_step_table_f32", *args, e.p("time_table"), TIME_TABLE_ROWS, e.stream)
            else:
                self._add("nd_cond_step_f32", *args, e.stream)
            return
        self._add("nd_sinusoidal_time_emb_f32", self.time.data_ptr(), e.p("time_freqs"), self.emb.data_ptr(), self.B, d // 2, e.stream)
        self.linear_rows(self.emb, e.p("time_mlp.1.weight"), e.p("time_mlp.1.bias"), self.t1, d, 4 * d, act_out=L.ACT_GELU)
        # every consumer applies SiLU first (ResnetBlock.mlp[0]), so it is applied once here
        self.linear_rows(self.t1, e.p("time_mlp.3.weight"), e.p("time_mlp.3.bias"), self.st, 4 * d, 4 * d, act_out=L.ACT_SILU)
        self.linear_rows(self.st, e.p("tproj.weight"), e.p("tproj.bias"), self.tproj, 4 * d, e.tproj_rows)

    def _record_net(self) -> None:
        e, B, H, W, d = self.e, self.B, self.H, self.W, self.e.dim
        G, tr = RESNET_GROUPS, self.e.traits
        shot_noise = None
        if tr.shot_branch:
            # ---- shot-noise branch, full resolution (:598-604)
            r_shot = self.mlp("shot_mlp1", self._src(self.clean, self.x), 2 * e.inp_dim, d, d, H, W)
            s = self.attn_block("shot_attn", r_shot, H, W)
            s2 = self.mlp("shot_mlp2", self._src(s), d, d, d, H, W)
            s3 = self.resnet("shot_time", s2, None, d, H, W, SHOT_GROUPS, extra_res=r_shot)   # shot_time(...) + r
            shot_noise = self._tap("shot_noise", self.mlp("shot_mlp3", self._src(s3), d, d, e.inp_dim, H, W))
            for nm, tt in (("shot_mlp1", r_shot), ("shot_attn", s), ("shot_mlp2", s2), ("shot_time", s3)):
                self._tap(nm, tt)
            self._release(r_shot, s, s2, s3)
        # ---- trunk
        x0 = self._alloc(B, H, W, d)
        stem_split = "init_conv.weight.split" in e.slots
        self._add("nd_conv7x7_c4_split_f32" if stem_split else "nd_conv7x7_c4_f32", self.x.data_ptr(),
                  e.p("init_conv.weight.split" if stem_split else "init_conv.weight"), e.p("init_conv.bias"), x0.data_ptr(), d, B, H, W, d, e.stream)
        self._tap("init_conv", x0)
        xin = x0
        if tr.cond_branch:          # x = cond_concat_conv(cat[init_conv(x), clean_emb])   others_arch.py:495-498
            xin, *_ = self.conv3("cond_concat_conv", self._src(x0, self.clean_emb), 2 * d, d, H, W, stats=False)
            self._tap("cond_concat", xin)
        pm1, pm2 = ((self.pos_e, self.pos_e) if self.posgen else (self.posmap1, self.posmap2)) if tr.position else (None, None)   # else plain ResnetBlocks without time
        x = self._tap("pos_block1", self.resnet("pos_block1", xin, None, d, H, W, POS_GROUPS, posmap=pm1))
        if xin is not x0:
            self._release(xin)
        rs = 3 if tr.iso_attn else 2            # stage index of the resampling layer
        hs: List[torch.Tensor] = []
        h, w = H, W
        for i, (cin, cout) in enumerate(stage_dims(d)):
            p = f"downs.{i}"
            x1 = self.resnet(p + ".0", x, None, cin, h, w, G)
            self._release(x)
            x2 = self.resnet(p + ".1", x1, None, cin, h, w, G)
            if e.stage_attn and e.stage_attn[i]:
                x2 = self._stage_attention(f"down_attns.{i}", e.stage_attn[i], x2, h, w)
            hs += [x1, x2]
            xa = self.attn_block(p + ".2", x2, h, w) if tr.iso_attn else x2
            self._tap(p + ".0", x1); self._tap(p + ".1", x2); self._tap(p + ".2", xa)
            if i == 3:
                x, *_ = self.conv3(f"{p}.{rs}", self._src(xa), cin, cout, h, w, stats=False)
            else:
                h, w = h // 2, w // 2
                x = self.pw(f"{p}.{rs}.1", self._src(xa, None, L.PRO_NONE, unshuffle=1, c0=4 * cin, ld0=cin), 4 * cin, cout,
                            h * w, w).view(B, h, w, cout)
            if xa is not x2:
                self._release(xa)
            self._tap(f"down{i}", x)
        mid = x.shape[-1]
        xm = self.resnet("mid_block1", x, None, mid, h, w, G)
        self._release(x)
        if e.mid_attn:
            xm = self._mid_attention(xm, h, w)
        x = self._tap("mid", self.resnet("mid_block2", xm, None, mid, h, w, G))
        self._release(xm)
        for i, (cin, cout) in enumerate(reversed(stage_dims(d))):
            p = f"ups.{i}"
            sk = hs.pop()
            x1 = self.resnet(p + ".0", x, sk, cout, h, w, G)
            self._release(x, sk)
            sk = hs.pop()
            x2 = self.resnet(p + ".1", x1, sk, cout, h, w, G)
            self._release(x1, sk)
            if e.stage_attn and e.stage_attn[3 - i]:
                x2 = self._stage_attention(f"up_attns.{i}", e.stage_attn[3 - i], x2, h, w)
            xa = self.attn_block(p + ".2", x2, h, w) if tr.iso_attn else x2
            if xa is not x2:
                self._release(x2)
            if i == 3:
                x, *_ = self.conv3(f"{p}.{rs}", self._src(xa), cout, cin, h, w, stats=False)
            else:
                h, w = h * 2, w * 2
                x, *_ = self.conv3(f"{p}.{rs}.1", self._src(xa, None, L.PRO_NONE, upsample=1), cout, cin, h, w, stats=False)
            self._release(xa)
            self._tap(p + ".0", x1); self._tap(p + ".1", x2); self._tap(p + ".2", xa); self._tap(f"up{i}", x)
        xp = self._tap("pos_block2", self.resnet("pos_block2", x, None, d, H, W, POS_GROUPS, posmap=pm2))
        self._release(x)
        xf = self._tap("final_res_block", self.resnet("final_res_block", xp, x0, d, H, W, G))
        self._release(xp, x0)
        # NoiseDiffNet: shot + read (:644); the ablation nets return final_conv(x) alone
        self.pw("final_conv", self._src(xf), d, e.inp_dim, H * W, W, res0=shot_noise, out=self.model_out)
        self._release(xf)
        if shot_noise is not None:
            self._release(shot_noise)

    def _mid_attention(self, x: torch.Tensor, h: int, w: int, prefix: str = "mid_attn") -> torch.Tensor:
        """x = Attention(x) + x (Diffusion_arch.py:237-266): between the mid blocks (BASELINE config 4) or as a stage's full attention."""
        e, B, Cc, N = self.e, self.B, x.shape[-1], h * w
        hid = ATTN_HEADS * ATTN_DIM_HEAD
        xn = self._alloc(B, N, Cc)
        self._add("nd_rmsnorm_nhwc_f32", x.data_ptr(), Cc, e.p(prefix + ".norm.g"), xn.data_ptr(), Cc, B, N, Cc, e.stream)
        qkv = self.pw(prefix + ".to_qkv", self._src(xn), Cc, 3 * hid, N, w, bias=False)
        att = self._alloc(B, N, hid)
        self._add("nd_attention_mfma_f32", qkv.data_ptr(), 3 * hid, att.data_ptr(), hid, B, N, ATTN_HEADS, ATTN_DIM_HEAD, e.stream)
        y = self.pw(prefix + ".to_out", self._src(att), hid, Cc, N, w, res0=x)
        self._release(xn, qkv, att, x)
        return y.view(B, h, w, Cc)

    def _stage_attention(self, prefix: str, kind: str, x: torch.Tensor, h: int, w: int) -> torch.Tensor:
        """x = attn(x) + x behind a stage's second ResnetBlock -- upstream's per-stage wiring of the classes the reference defines and drops
        (Diffusion_arch.py:198-266,509-518): 'full' = Attention (as _mid_attention), 'linear' = LinearAttention: RMSNorm ->
        to_qkv -> softmax(q over d), softmax(k over n), (k v^T)^T q (nd_linear_attention_f32) -> to_out.0 -> RMSNorm (+ x)."""
        if kind == "full":
            return self._mid_attention(x, h, w, prefix)
        e, B, Cc, N = self.e, self.B, x.shape[-1], h * w
        hid = ATTN_HEADS * ATTN_DIM_HEAD
        xn = self._alloc(B, N, Cc)
        self._add("nd_rmsnorm_nhwc_f32", x.data_ptr(), Cc, e.p(prefix + ".norm.g"), xn.data_ptr(), Cc, B, N, Cc, e.stream)
        qkv = self.pw(prefix + ".to_qkv", self._src(xn), Cc, 3 * hid, N, w, bias=False)
        att = self._alloc(B, N, hid)
        ws = self._alloc(int(e.lib.nd_linear_attention_workspace_floats(B, N, ATTN_HEADS)))
        self._add("nd_linear_attention_f32", qkv.data_ptr(), 3 * hid, att.data_ptr(), hid, ws.data_ptr(), B, N, ATTN_HEADS, ATTN_DIM_HEAD, e.stream)
        o = self.pw(prefix + ".to_out.0", self._src(att), hid, Cc, N, w)
        y = self._alloc(B, N, Cc)
        self._add("nd_rmsnorm_add_nhwc_f32", o.data_ptr(), Cc, e.p(prefix + ".to_out.1.g"), x.data_ptr(), Cc, y.data_ptr(), Cc, B, N, Cc, e.stream)
        self._release(xn, qkv, att, ws, o, x)
        return y.view(B, h, w, Cc)

    # ------------------------------------------------------------------ execution
    @staticmethod
    def run(ops: List[Op]) -> None:
        for fn, args, name, _meta in ops:
            r = fn(*args)
            if r != 0:
                L.check(r, name)

    def set_condition(self, condition) -> None:
        """Upload clean_img / position / iso_ratio_idx and run the step-invariant part once.  Nets without the position
        or ISO inputs ignore those keys; UNet_PosEmbV2_NoPosition also takes the bare clean image (others_arch.py:658)."""
        B, H, W, e = self.B, self.H, self.W, self.e
        tr = e.traits
        if torch.is_tensor(condition):
            if tr.position or tr.iso_attn:
                raise TypeError(f"{e.arch} needs the condition dict (clean_img, position, iso_ratio_idx), got a tensor")
            condition = {"clean_img": condition}
        clean = condition["clean_img"]
        if tuple(clean.shape) != (B, e.inp_dim, H, W):
            raise ValueError(f"condition shapes: clean_img {tuple(clean.shape)} do not match batch {(B, e.inp_dim, H, W)}")
        with torch.cuda.device(self.dev):
            self.nchw_tmp.copy_(clean.to(self.dev, torch.float32))
            if tr.position:
                pos = condition["position"]
                if tuple(pos.shape) != (B, 2, H, W):
                    raise ValueError(f"condition shapes: position {tuple(pos.shape)} do not match batch {(B, 2, H, W)}")
                self.position.copy_(pos.to(self.dev, torch.float32))
            if tr.iso_attn:
                iso = condition["iso_ratio_idx"]
                if tuple(iso.shape) != (B,):
                    raise ValueError(f"condition shapes: iso_ratio_idx {tuple(iso.shape)} do not match batch {(B,)}")
                iso = iso.to(torch.int64)
                if int(iso.min()) < 0 or int(iso.max()) >= ISO_TABLE_ROWS:
                    raise IndexError("iso_ratio_idx out of range for nn.Embedding(100, 16)")
                self.iso_idx.copy_(iso.to(self.dev))    # the reference leaves it on the CPU (trainer_diffusion.py:135)
            torch.cuda.synchronize(self.dev)
            L.call("nd_nchw_to_nhwc_f32", self.nchw_tmp.data_ptr(), self.clean.data_ptr(), B, e.inp_dim, H, W, e.stream)
            self.run(self.cond_ops)
            e.sync()
        self.condition_set = True

    def load_x(self, x_nchw: torch.Tensor) -> None:
        e = self.e
        with torch.cuda.device(self.dev):
            self.nchw_tmp.copy_(x_nchw.to(self.dev, torch.float32))
            torch.cuda.synchronize(self.dev)
            L.call("nd_nchw_to_nhwc_f32", self.nchw_tmp.data_ptr(), self.x.data_ptr(), self.B, e.inp_dim, self.H, self.W, e.stream)

    def read_nchw(self, nhwc: torch.Tensor) -> torch.Tensor:
        e = self.e
        with torch.cuda.device(self.dev), torch.inference_mode(False):
            out = torch.empty(self.B, e.inp_dim, self.H, self.W, dtype=torch.float32, device=self.dev)
            L.call("nd_nhwc_to_nchw_f32", nhwc.data_ptr(), out.data_ptr(), self.B, e.inp_dim, self.H, self.W, e.stream)
            e.sync()
        return out

    def forward(self, x_nchw: torch.Tensor, time: torch.Tensor) -> torch.Tensor:
        """One NoiseDiffNet.forward on already-set conditions; returns NCHW."""
        if not self.condition_set:
            raise L.HipError("Plan.forward: set_condition() first")
        with torch.cuda.device(self.dev):
            self.time.copy_(time.to(self.dev, torch.int64))
        self.load_x(x_nchw)
        self.run(self.step_ops)
        return self.read_nchw(self.model_out)
