"""Parameter and layer inventory of NoiseDiffNet, as data.

The drop-in contract includes the network's state-dict: 416
tensors whose names and shapes must equal those produced by the reference
constructor (models/archs/Diffusion_arch.py:447-570), so that
``pretrained_ckpts/DiffusionNet_ckpt.pth`` loads with ``strict=True``
(models/trainer_diffusion.py:333-349).  Instead of re-declaring a torch module
tree, this file enumerates the tensors in registration order; ``net.py`` hangs
them on a generic container, ``engine.py`` packs them into the device arena and
``synth.py`` fills them.  ``tests/golden/state_dict_keys_*.json`` (captured from
the reference) pins the list.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple, Optional, Sequence

ISO_TABLE_ROWS = 100      # nn.Embedding(100, 16)      Diffusion_arch.py:486-487
ISO_DIM = 16
POS_DIM = 8               # Diffusion_arch.py:473
ATTN_HEADS = 4            # AttnBlock(heads=4, dim_head=32)  :532,546,567
ATTN_DIM_HEAD = 32
DIM_MULTS = (1, 2, 4, 8)  # :457
RESNET_GROUPS = 8         # :459
POS_GROUPS = 2            # :560
SHOT_GROUPS = 2           # :569


@dataclass(frozen=True)
class ParamSpec:
    name: str
    shape: Tuple[int, ...]
    init: str          # uniform_fan_in | ones | zeros | normal
    fan_in: int = 0
    std: float = 1.0   # standard deviation for init == "normal"


class _Lister:
    def __init__(self) -> None:
        self.items: List[ParamSpec] = []

    def conv(self, name: str, cin: int, cout: int, k: int, bias: bool = True) -> None:
        fan = cin * k * k
        self.items.append(ParamSpec(f"{name}.weight", (cout, cin, k, k), "uniform_fan_in", fan))
        if bias:
            self.items.append(ParamSpec(f"{name}.bias", (cout,), "uniform_fan_in", fan))

    def linear(self, name: str, cin: int, cout: int, bias: bool = True) -> None:
        self.items.append(ParamSpec(f"{name}.weight", (cout, cin), "uniform_fan_in", cin))
        if bias:
            self.items.append(ParamSpec(f"{name}.bias", (cout,), "uniform_fan_in", cin))

    def norm(self, name: str, c: int) -> None:
        self.items.append(ParamSpec(f"{name}.weight", (c,), "ones"))
        self.items.append(ParamSpec(f"{name}.bias", (c,), "zeros"))

    # composite layers, in the reference's registration order -----------------
    def resnet(self, name: str, cin: int, cout: int, time_dim: int) -> None:
        # ResnetBlock: mlp(SiLU, Linear), block1, block2, res_conv   :146-156
        self.linear(f"{name}.mlp.1", time_dim, cout * 2)
        self._blocks(name, cin, cout)

    def resnet_pos(self, name: str, cin: int, cout: int, pos_dim: int) -> None:
        # ResnetBlock2: mlp(SiLU, Conv1x1), block1, block2, res_conv  :173-183
        self.conv(f"{name}.mlp.1", pos_dim, cout * 2, 1)
        self._blocks(name, cin, cout)

    def _blocks(self, name: str, cin: int, cout: int) -> None:
        self.conv(f"{name}.block1.proj", cin, cout, 3)     # hard-wired 3x3  :154
        self.norm(f"{name}.block1.norm", cout)
        self.conv(f"{name}.block2.proj", cout, cout, 3)
        self.norm(f"{name}.block2.norm", cout)
        if cin != cout:
            self.conv(f"{name}.res_conv", cin, cout, 1)

    def attn_block(self, name: str, c: int) -> None:
        # AttnBlock: attn(to_q,to_k,to_v,to_out.0), norm1, norm2, ff, proj_out  :425-432
        inner = ATTN_HEADS * ATTN_DIM_HEAD
        self.linear(f"{name}.attn.to_q", c, inner, bias=False)
        self.linear(f"{name}.attn.to_k", ISO_DIM, inner, bias=False)
        self.linear(f"{name}.attn.to_v", ISO_DIM, inner, bias=False)
        self.linear(f"{name}.attn.to_out.0", inner, c)
        self.norm(f"{name}.norm1", c)
        self.norm(f"{name}.norm2", c)
        self.linear(f"{name}.ff.net.0.0", c, 2 * c)
        self.linear(f"{name}.ff.net.2", 2 * c, c)
        self.conv(f"{name}.proj_out", c, c, 1)

    def mlp(self, name: str, cin: int, hidden: int, cout: int) -> None:
        self.conv(f"{name}.fc1", cin, hidden, 1)
        self.conv(f"{name}.fc2", hidden, cout, 1)


def stage_dims(dim: int) -> List[Tuple[int, int]]:
    dims = [dim] + [dim * m for m in DIM_MULTS]          # :480
    return list(zip(dims[:-1], dims[1:]))                # :481


def noisediff_param_spec(dim: int, inp_dim: int = 4) -> List[ParamSpec]:
    """All state-dict tensors of ``NoiseDiffNet(args)`` with ``args.dim = dim``."""
    L = _Lister()
    time_dim = dim * 4
    in_out = stage_dims(dim)
    n_res = len(in_out)

    L.conv("init_conv", inp_dim, dim, 7)                                  # :478
    L.items.append(ParamSpec("iso_embed.weight", (ISO_TABLE_ROWS, ISO_DIM), "normal"))
    L.linear("time_mlp.1", dim, time_dim)                                 # :502-507
    L.linear("time_mlp.3", time_dim, time_dim)

    for i, (cin, cout) in enumerate(in_out):                              # :526-534
        last = i >= n_res - 1
        L.resnet(f"downs.{i}.0", cin, cin, time_dim)
        L.resnet(f"downs.{i}.1", cin, cin, time_dim)
        L.attn_block(f"downs.{i}.2", cin)
        if last:
            L.conv(f"downs.{i}.3", cin, cout, 3)
        else:
            L.conv(f"downs.{i}.3.1", cin * 4, cout, 1)                    # :78-82

    for i, (cin, cout) in enumerate(reversed(in_out)):                    # :540-548
        last = i == n_res - 1
        L.resnet(f"ups.{i}.0", cout + cin, cout, time_dim)
        L.resnet(f"ups.{i}.1", cout + cin, cout, time_dim)
        L.attn_block(f"ups.{i}.2", cout)
        if last:
            L.conv(f"ups.{i}.3", cout, cin, 3)
        else:
            L.conv(f"ups.{i}.3.1", cout, cin, 3)                          # :72-76

    # self.downs / self.ups are both registered (:522-523) before the mid blocks (:537-538)
    mid = in_out[-1][1]
    L.resnet("mid_block1", mid, mid, time_dim)
    L.resnet("mid_block2", mid, mid, time_dim)

    L.resnet("final_res_block", dim * 2, dim, time_dim)                   # :553
    L.conv("final_conv", dim, inp_dim, 1)                                 # :554

    L.conv("pos_enc.weights", 2, POS_DIM, 1)                              # :558, :328
    L.mlp("pos_mlp", POS_DIM * 3, POS_DIM * 2, POS_DIM)                   # :559
    L.resnet_pos("pos_block1", dim, dim, POS_DIM)                         # :561
    L.resnet_pos("pos_block2", dim, dim, POS_DIM)                         # :562

    L.mlp("shot_mlp1", inp_dim * 2, dim, dim)                             # :566
    L.attn_block("shot_attn", dim)                                        # :567
    L.mlp("shot_mlp2", dim, dim, dim)                                     # :568
    L.resnet("shot_time", dim, dim, time_dim)                             # :569 (3x3, see quirk)
    L.mlp("shot_mlp3", dim, dim, inp_dim)                                 # :570
    return L.items


ARCHS = ("NoiseDiffNet", "UNet_PosEmbV2", "UNet_PosEmbV2_NoPosition", "UNet_PosEmbV2_CameraCond")


@dataclass(frozen=True)
class ArchTraits:
    """What distinguishes the four U-Nets the reference ships (Diffusion_arch.py:447-646, others_arch.py:364-985)."""
    shot_branch: bool      # shot_mlp1..3 / shot_attn / shot_time, output = shot + read   (NoiseDiffNet only)
    iso_attn: bool         # AttnBlock(ISO context) after every stage's two ResnetBlocks
    position: bool         # pos_enc + pos_mlp, pos_block1/2 are ResnetBlock2 (else plain ResnetBlocks without time)
    cond_branch: bool      # cond_init_conv 7x7 -> cond_res_block1 -> cond_concat_conv on cat[init_conv(x), clean_emb]


def arch_traits(arch: str) -> ArchTraits:
    if arch == "NoiseDiffNet":
        return ArchTraits(True, True, True, False)
    if arch == "UNet_PosEmbV2":                 # others_arch.py:364-537
        return ArchTraits(False, False, True, True)
    if arch == "UNet_PosEmbV2_NoPosition":      # others_arch.py:540-707
        return ArchTraits(False, False, False, True)
    if arch == "UNet_PosEmbV2_CameraCond":      # others_arch.py:796-985
        return ArchTraits(False, True, True, True)
    raise KeyError(f"unknown arch {arch!r}; this build has {ARCHS}")


def posemb_unet_param_spec(arch: str, dim: int, inp_dim: int = 4, cond_dim: int = 4) -> List[ParamSpec]:
    """State-dict tensors of the ``UNet_PosEmbV2*`` ablation nets (others_arch.py:364-985) in registration order."""
    tr = arch_traits(arch)
    assert tr.cond_branch, arch
    L = _Lister()
    time_dim = dim * 4
    in_out = stage_dims(dim)
    n_res = len(in_out)
    rs = 3 if tr.iso_attn else 2                                           # index of the resampling layer in a stage

    L.conv("init_conv", inp_dim, dim, 7)                                  # :394 / :570 / :826
    if tr.iso_attn:
        L.items.append(ParamSpec("iso_embed.weight", (ISO_TABLE_ROWS, ISO_DIM), "normal"))   # :833-834
    L.linear("time_mlp.1", dim, time_dim)
    L.linear("time_mlp.3", time_dim, time_dim)
    for i, (cin, cout) in enumerate(in_out):                              # :437-444 / :869-877
        L.resnet(f"downs.{i}.0", cin, cin, time_dim)
        L.resnet(f"downs.{i}.1", cin, cin, time_dim)
        if tr.iso_attn:
            L.attn_block(f"downs.{i}.2", cin)
        if i >= n_res - 1:
            L.conv(f"downs.{i}.{rs}", cin, cout, 3)
        else:
            L.conv(f"downs.{i}.{rs}.1", cin * 4, cout, 1)
    for i, (cin, cout) in enumerate(reversed(in_out)):                    # :450-457 / :883-891
        L.resnet(f"ups.{i}.0", cout + cin, cout, time_dim)
        L.resnet(f"ups.{i}.1", cout + cin, cout, time_dim)
        if tr.iso_attn:
            L.attn_block(f"ups.{i}.2", cout)
        if i == n_res - 1:
            L.conv(f"ups.{i}.{rs}", cout, cin, 3)
        else:
            L.conv(f"ups.{i}.{rs}.1", cout, cin, 3)
    mid = in_out[-1][1]
    L.resnet("mid_block1", mid, mid, time_dim)
    L.resnet("mid_block2", mid, mid, time_dim)
    L.resnet("final_res_block", dim * 2, dim, time_dim)
    L.conv("final_conv", dim, inp_dim, 1)
    if tr.position:
        L.conv("pos_enc.weights", 2, POS_DIM, 1)                          # :467-471
        L.mlp("pos_mlp", POS_DIM * 3, POS_DIM * 2, POS_DIM)
        L.resnet_pos("pos_block1", dim, dim, POS_DIM)
        L.resnet_pos("pos_block2", dim, dim, POS_DIM)
    else:
        L._blocks("pos_block1", dim, dim)                                 # ResnetBlock(time_emb_dim=None, groups=2)  :644-646
        L._blocks("pos_block2", dim, dim)
    L.conv("cond_init_conv", cond_dim, dim, 7)                            # :474-477
    L._blocks("cond_res_block1", dim, dim)                                # ResnetBlock(time_emb_dim=None, groups=8)
    L.conv("cond_concat_conv", dim * 2, dim, 3)
    return L.items


def arch_param_spec(arch: str, dim: int, inp_dim: int = 4) -> List[ParamSpec]:
    return noisediff_param_spec(dim, inp_dim) if arch == "NoiseDiffNet" else posemb_unet_param_spec(arch, dim, inp_dim)


def attention_param_spec(prefix: str, dim: int, heads: int = 4, dim_head: int = 32) -> List[ParamSpec]:
    """Standalone ``Attention`` block (Diffusion_arch.py:237-253), config-4 extension."""
    hidden = heads * dim_head
    return [
        ParamSpec(f"{prefix}.norm.g", (1, dim, 1, 1), "ones"),
        ParamSpec(f"{prefix}.to_qkv.weight", (hidden * 3, dim, 1, 1), "uniform_fan_in", dim),
        ParamSpec(f"{prefix}.to_out.weight", (dim, hidden, 1, 1), "uniform_fan_in", hidden),
        ParamSpec(f"{prefix}.to_out.bias", (dim,), "uniform_fan_in", hidden),
    ]


def linear_attention_param_spec(prefix: str, dim: int, heads: int = 4, dim_head: int = 32) -> List[ParamSpec]:
    """Standalone ``LinearAttention`` block (Diffusion_arch.py:198-216): RMSNorm, to_qkv (no bias), to_out = conv1x1 + RMSNorm."""
    hidden = heads * dim_head
    return [
        ParamSpec(f"{prefix}.norm.g", (1, dim, 1, 1), "ones"),
        ParamSpec(f"{prefix}.to_qkv.weight", (hidden * 3, dim, 1, 1), "uniform_fan_in", dim),
        ParamSpec(f"{prefix}.to_out.0.weight", (dim, hidden, 1, 1), "uniform_fan_in", hidden),
        ParamSpec(f"{prefix}.to_out.0.bias", (dim,), "uniform_fan_in", hidden),
        ParamSpec(f"{prefix}.to_out.1.g", (1, dim, 1, 1), "ones"),
    ]


# The reference computes ``full_attn = (False, False, False, True)`` and ``FullAttention`` per stage (Diffusion_arch.py:467-468,509-518) and then
# drops them; upstream (lucidrains' Unet) wires ``attn_klass = FullAttention if full_attn else LinearAttention`` as the third module of every
# down and up stage, applied as ``x = attn(x) + x`` behind the stage's second ResnetBlock and in front of the skip.  ``stage_attn`` switches that
# wiring on (second half): None = the reference network as it runs; True = the reference's own tuple.
STAGE_ATTN_REFERENCE = ("linear", "linear", "linear", "full")


def normalize_stage_attn(value) -> Optional[Tuple[Optional[str], ...]]:
    """None / False -> None; True -> the reference's ``full_attn`` tuple; a 4-tuple of 'linear' / 'full' / None, or of bools with upstream's
    meaning (False = LinearAttention, True = full Attention)."""
    if value is None or value is False:
        return None
    if value is True:
        return STAGE_ATTN_REFERENCE
    kinds = tuple(("full" if v else "linear") if isinstance(v, bool) else v for v in value)
    if len(kinds) != len(DIM_MULTS) or any(k not in (None, "linear", "full") for k in kinds):
        raise ValueError(f"stage_attn={value!r}: expected True or {len(DIM_MULTS)} entries of 'linear' / 'full' / None (or upstream's full_attn booleans)")
    return kinds if any(kinds) else None


def stage_attention_param_spec(dim: int, kinds: Sequence[Optional[str]]) -> List[ParamSpec]:
    """Parameters of the per-stage attention modules: ``down_attns.{i}`` at the stage's input width, ``up_attns.{i}`` (i = 0 the deepest) at
    its ResnetBlocks' output width -- upstream's ``attn_klass(dim_in)`` / ``attn_klass(dim_out)``."""
    items: List[ParamSpec] = []
    dims = stage_dims(dim)
    for i, k in enumerate(kinds):
        if k:
            items += (attention_param_spec if k == "full" else linear_attention_param_spec)(f"down_attns.{i}", dims[i][0])
    for i, (k, (_, cout)) in enumerate(zip(reversed(tuple(kinds)), reversed(dims))):
        if k:
            items += (attention_param_spec if k == "full" else linear_attention_param_spec)(f"up_attns.{i}", cout)
    return items


LSID_STAGES = (32, 64, 128, 256, 512)      # models/archs/SID_arch.py:57-75


def lsid_param_spec(inchannel: int = 4) -> List[ParamSpec]:
    """State-dict tensors of ``LSID(args)`` (models/archs/SID_arch.py:49-103) in registration order.
    Init as upstream: Conv2d / ConvTranspose2d weights ~ N(0, sqrt(2 / (k*k*out_channels))), biases 0 (:96-103)."""
    import math
    items: List[ParamSpec] = []

    def conv(name, cin, cout, k):
        items.append(ParamSpec(f"{name}.weight", (cout, cin, k, k), "normal", std=math.sqrt(2.0 / (k * k * cout))))
        items.append(ParamSpec(f"{name}.bias", (cout,), "zeros"))

    def up(name, cin, cout):       # ConvTranspose2d weight is (in, out, kH, kW); no bias (:77)
        items.append(ParamSpec(f"{name}.weight", (cin, cout, 2, 2), "normal", std=math.sqrt(2.0 / (4 * cout))))

    prev = inchannel
    for i, c in enumerate(LSID_STAGES, start=1):
        conv(f"conv{i}_1", prev, c, 3)
        conv(f"conv{i}_2", c, c, 3)
        prev = c
    for i, c in zip(range(6, 10), reversed(LSID_STAGES[:-1])):
        up(f"up{i}", prev, c)
        conv(f"conv{i}_1", 2 * c, c, 3)
        conv(f"conv{i}_2", c, c, 3)
        prev = c
    conv("conv10", prev, inchannel, 1)
    return items
