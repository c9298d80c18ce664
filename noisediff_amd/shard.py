"""Data-parallel sampling across the GPUs of one node: one process per GPU, no per-step collectives.

Every sample of a batch is independent through all T steps (GroupNorm / LayerNorm / attention are
per-sample; there is no BatchNorm), so rank r of R simply takes rows [lo, hi) of the global batch
(SURVEY.md 8e).  The reference instead wraps the net in nn.DataParallel, which re-broadcasts all
weights and scatters/gathers activations on EVERY step (models/modules.py:81,
models/denoising_diffusion_pytorch.py:332).  Here the only data-path collective is ONE broadcast
of the packed weight arena at start-up (RCCL over xGMI; ``backend='nccl'`` is RCCL on ROCm), plus an
optional all-gather of the finished patches.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous rows [lo, hi) of rank `rank`; the remainder goes to the first ranks."""
    if not (0 <= rank < world) or total < 0:
        raise ValueError(f"bad shard request total={total} rank={rank} world={world}")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_weights(net, device: torch.device, src: int = 0, group=None, shapes: Optional[Sequence[Tuple[int, int, int]]] = None,
                      packed: bool = False):
    """Fill this rank's engine from rank `src`'s weights: the one collective of the data path.

    Default: the network's own fp32 weights travel as one flat buffer (150 MB at d=64) and every rank packs its arena itself
    (``Engine.broadcast_state_dict``).  ``packed=True`` ships rank `src`'s packed arena instead (0.9 GB at d=64: it holds up to three
    packings of every 3x3 weight); with ``shapes`` -- the (batch, height, width) problems this job will run -- only the slices
    those plans read (0.5 GB), and another problem size on a receiving rank then raises instead of reading weights that never arrived."""
    from .engine import Engine
    eng = Engine(net.dim, device, mid_attn=net.has_mid_attn, inp_dim=net.channels, arch=getattr(net, "ARCH", "NoiseDiffNet"),
                 stage_attn=getattr(net, "stage_attn", None))
    is_src = dist.get_rank(group) == src
    if not packed:
        eng.last_broadcast_bytes = eng.broadcast_state_dict(net.state_dict() if is_src else None, src=src, group=group)
    else:
        if is_src:
            eng.load_state_dict(net.state_dict())
        for B, H, W in shapes or ():
            eng.plan(B, H, W, allow_empty=True)
        eng.last_broadcast_bytes = eng.broadcast(src=src, group=group, only_used=bool(shapes))
    net.adopt_engine(eng)
    return eng


def sample_sharded(sample_fn: Callable[..., torch.Tensor], total_batch: int,
                   make_condition: Callable[[int, int], Dict[str, torch.Tensor]], *, seed: int,
                   set_offset: Optional[Callable[[int], None]] = None, gather: bool = False, group=None,
                   device: Optional[torch.device] = None, **kw):
    """Run ``sample_fn(batch_size=hi-lo, condition=make_condition(lo, hi), seed=seed, **kw)`` on this
    rank's rows.  ``set_offset(lo)`` tells the sampler the global index of its first sample so that the
    device noise stream of row i is the same whatever the number of ranks.  With ``gather`` every rank
    returns the full (total_batch, ...) result (one all-gather at the very end), else its own rows.
    ``device``: where a rank WITHOUT rows (world > total_batch) builds its part of the all-gather -- the collective's backend decides
    what it accepts (RCCL: device tensors only), so the placeholder must live where the other ranks' results live; default: this
    rank's current CUDA device under the nccl backend, else the CPU."""
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)
    if total_batch <= 0:
        raise ValueError(f"sample_sharded: total_batch={total_batch}: nothing to sample")
    lo, hi = shard_bounds(total_batch, rank, world)
    if set_offset is not None:
        set_offset(lo)
    out = sample_fn(batch_size=hi - lo, condition=make_condition(lo, hi), seed=seed, **kw) if hi > lo else None
    if not gather or world == 1:
        return out
    sizes = [shard_bounds(total_batch, r, world) for r in range(world)]
    n_max = max(h - l for l, h in sizes)
    tail = tuple(out.shape[1:]) if out is not None else None
    shapes = [None] * world
    dist.all_gather_object(shapes, tail, group=group)
    tail = next(s for s in shapes if s is not None)
    if out is not None:
        pad_dev = out.device
    elif device is not None:
        pad_dev = torch.device(device)
    else:       # a rank without rows: the all-gather of an RCCL group takes device tensors only (r3 built this placeholder on the CPU)
        pad_dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    pad = torch.zeros((n_max,) + tail, dtype=out.dtype if out is not None else torch.float32, device=pad_dev)
    if out is not None:
        pad[: hi - lo] = out
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[: h - l] for p, (l, h) in zip(parts, sizes)], dim=0)
