"""Deterministic synthetic data: weights, conditions and noise without any files.

There is no network and no SID RAW data on the GPU box, and checkpoints are too
large to commit, so every tensor the sampler needs (the 416 state-dict tensors,
``clean_img`` / ``position`` / ``iso_ratio_idx`` conditions, x_T and per-step
noise in parity mode) is generated from a counter-based hash keyed by
``(seed, name, flat index)``.  The same function is used by the golden-capture
script (which loads the result into the reference via ``load_state_dict``), by
the tests and by ``bench.py`` -- so a tensor named ``downs.0.0.block1.proj.weight``
with seed 0 is bit-identical everywhere, on any box, independent of torch's RNG.

Distribution choices follow the reference's *effective* initialisation: it never
calls ``init_weights`` (models/modules.py:82 is commented out), so Conv2d/Linear
keep PyTorch's default U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias,
norm layers are (1, 0) and ``nn.Embedding`` is N(0, 1).
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Sequence, Tuple

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _fnv1a64(text: str) -> int:
    h = 0xCBF29CE484222325
    for ch in text.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(z: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = (z + _GOLD) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def _stream_base(seed: int, name: str) -> np.uint64:
    mixed = (int(seed) * 0xD1342543DE82EF95 + _fnv1a64(name)) & 0xFFFFFFFFFFFFFFFF
    return _splitmix64(np.array([mixed], dtype=np.uint64))[0]


def raw_u64(seed: int, name: str, count: int, offset: int = 0) -> np.ndarray:
    """``count`` hash words for stream (seed, name), starting at element ``offset``."""
    idx = np.arange(offset, offset + count, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return _splitmix64((idx * _GOLD + _stream_base(seed, name)) & _MASK)


def uniform01(seed: int, name: str, count: int, offset: int = 0) -> np.ndarray:
    """float64 uniforms in [0, 1) with 53 random bits."""
    return (raw_u64(seed, name, count, offset) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def uniform(seed: int, name: str, shape: Sequence[int], lo: float, hi: float) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(seed, name, n)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32).reshape(tuple(shape)))


def normal(seed: int, name: str, shape: Sequence[int], offset: int = 0) -> torch.Tensor:
    """Standard normals by Box-Muller on two independent hash streams.

    ``offset`` is in elements: ``normal(s, n, (B, ...))[b]`` equals
    ``normal(s, n, (...), offset=b * per_sample)`` -- which is what makes a rank's
    batch shard of the noise identical to the corresponding rows of the full batch.
    """
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = uniform01(seed, name + "#r", n, offset)
    u2 = uniform01(seed, name + "#a", n, offset)
    r = np.sqrt(-2.0 * np.log1p(-u1))          # 1-u1 in (0, 1]
    z = r * np.cos(2.0 * math.pi * u2)
    return torch.from_numpy(z.astype(np.float32).reshape(tuple(shape)))


def randint(seed: int, name: str, shape: Sequence[int], lo: int, hi: int) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    v = raw_u64(seed, name, n) % np.uint64(hi - lo)
    return torch.from_numpy((v.astype(np.int64) + lo).reshape(tuple(shape)))


def make_state_dict(spec: Iterable, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Fill every tensor of a parameter spec (see ``spec.noisediff_param_spec``)."""
    out: Dict[str, torch.Tensor] = {}
    for p in spec:
        if p.init == "uniform_fan_in":
            bound = 1.0 / math.sqrt(p.fan_in)
            out[p.name] = uniform(seed, p.name, p.shape, -bound, bound)
        elif p.init == "ones":
            out[p.name] = torch.ones(p.shape, dtype=torch.float32)
        elif p.init == "zeros":
            out[p.name] = torch.zeros(p.shape, dtype=torch.float32)
        elif p.init == "normal":
            out[p.name] = normal(seed, p.name, p.shape) * float(getattr(p, "std", 1.0))
        else:  # pragma: no cover - spec bug
            raise ValueError(f"unknown init kind {p.init!r} for {p.name}")
    return out


def make_position(batch: int, size: int, seed: int = 1,
                  frame_hw: Tuple[int, int] = (1424, 2128)) -> torch.Tensor:
    """(B, 2, H, W) normalised (row, col) coordinates of patches cut from the packed frame.

    Follows the sampling-time condition provider: ``make_coord(h, w, rescale=True)``
    (utils/util.py:138-147) gives coord[..., 0] = row/(h-1), coord[..., 1] = col/(w-1);
    patches are cropped on a regular grid (dataloader/dataset.py:203-219).  Patch
    origins here are drawn on the stride = size - size/4 grid from the hash stream.
    """
    fh, fw = frame_hw
    stride = max(size - size // 4, 1)
    ny = max((fh - size) // stride + 1, 1)
    nx = max((fw - size) // stride + 1, 1)
    oy = randint(seed, "position.oy", (batch,), 0, ny) * stride
    ox = randint(seed, "position.ox", (batch,), 0, nx) * stride
    rows = torch.arange(size, dtype=torch.float32)
    pos = torch.empty(batch, 2, size, size, dtype=torch.float32)
    for b in range(batch):
        r = (rows + float(oy[b])) / float(fh - 1)
        c = (rows + float(ox[b])) / float(fw - 1)
        pos[b, 0] = r[:, None].expand(size, size)
        pos[b, 1] = c[None, :].expand(size, size)
    return pos


def make_condition(batch: int, size: int, seed: int = 1, channels: int = 4,
                   first_sample: int = 0, total: int | None = None) -> Dict[str, torch.Tensor]:
    """Synthetic sampling condition dict with the shapes/dtypes of Trainer.test().

    clean_img ~ U[0,1) packed RGGB (B,4,H,W) fp32; position as above; iso_ratio_idx
    int64 in [0, 75) (75 entries in dataloader/combination_mapping.pickle).
    ``first_sample``/``total`` select rows [first, first+batch) of a ``total``-row
    global batch so that rank shards reproduce the full-batch tensors exactly.
    """
    total = batch + first_sample if total is None else total
    per = channels * size * size
    clean = torch.from_numpy(
        uniform01(seed, "clean_img", batch * per, first_sample * per)
        .astype(np.float32).reshape(batch, channels, size, size))
    pos = make_position(total, size, seed)[first_sample:first_sample + batch].contiguous()
    iso = randint(seed, "iso_ratio_idx", (total,), 0, 75)[first_sample:first_sample + batch].contiguous()
    return {"clean_img": clean, "position": pos, "iso_ratio_idx": iso}


def make_noise(seed: int, name: str, batch: int, channels: int, size: int,
               first_sample: int = 0) -> torch.Tensor:
    per = channels * size * size
    return normal(seed, name, (batch, channels, size, size), offset=first_sample * per)
