"""On-disk contract of the generated noise patches and the synth -> denoise composition.

The sampling driver of the reference writes one ``.npy`` per patch,
``<clean>+<noisy|clean>+<x>_<y>.npy`` holding a float32 ``(4, H, W)`` CHW array
(models/trainer_diffusion.py:296-325); the denoiser's dataset reads it back by splitting the name on
``'+'`` and the coordinate on ``'_'`` and forms its training input as
``clip(clip(noise, -1, 1) + clean, 0, 1)`` (dataloader/dataset_denoising.py:55-61,135-151).  This module
restates exactly that contract so the 8-GPU synthesizer can feed the reference's denoiser training
unchanged.  Plain numpy / torch host code: none of it is on the timed hot path.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

PACKED_W, PACKED_H = 4256 // 2, 2848 // 2      # packed Bayer frame of the SID Sony set (dataset.py:203)


def patch_grid(crop_size: int, w: int = PACKED_W, h: int = PACKED_H) -> List[Tuple[int, int]]:
    """(x, y) patch origins, row-major: stride = crop - crop/4, last patch flush with the border
    (dataloader/dataset.py:203-219).  512 -> 6 x 4 = 24 patches per frame."""
    ps = crop_size
    step = ps - ps // 4
    ys = list(range(0, h - ps + 1, step))
    if h - (ys[-1] + ps) < ps:
        ys.append(h - ps)
    xs = list(range(0, w - ps + 1, step))
    if w - (xs[-1] + ps) < ps:
        xs.append(w - ps)
    return [(x, y) for y in ys for x in xs]


def image_coord(x: int, y: int) -> str:
    return f"{int(x)}_{int(y)}"                               # dataset.py:273


def generated_name(clean_name: str, noisy_name: Optional[str], coord: str) -> str:
    """trainer_diffusion.py:310-316 (non-dark-frame branch)."""
    clean = clean_name.split(".ARW")[0]
    other = (noisy_name if noisy_name is not None else clean_name).split(".ARW")[0]
    return f"{clean}+{other}+{coord}.npy"


def parse_generated_name(path: str) -> Tuple[str, str, int, int]:
    """dataset_denoising.py:59-61,135-137: name.split('+') then coord.split('_') -> (clean, noisy, x, y)."""
    name = os.path.basename(path).split(".npy")[0]
    clean, noisy, coord = name.split("+")
    x, y = coord.split("_")
    return clean, noisy, int(x), int(y)


def save_generated(folder: str, output: torch.Tensor, clean_names: Sequence[str],
                   noisy_names: Optional[Sequence[str]], coords: Sequence[str]) -> List[str]:
    """Write a batch of sampled patches, one float32 (4, H, W) file each, into ``folder/generated``."""
    out_dir = os.path.join(folder, "generated")
    os.makedirs(out_dir, exist_ok=True)
    arr = output.detach().to("cpu", torch.float32).numpy()
    if arr.ndim != 4 or arr.shape[0] != len(coords):
        raise ValueError(f"expected (B, C, H, W) with B == {len(coords)}; got {arr.shape}")
    paths = []
    for i in range(arr.shape[0]):
        name = generated_name(clean_names[i], None if noisy_names is None else noisy_names[i], coords[i])
        path = os.path.join(out_dir, name)
        np.save(path, arr[i])
        paths.append(path)
    return paths


def compose_noisy(noise: torch.Tensor, clean: torch.Tensor) -> torch.Tensor:
    """noisy = clip(clip(noise, -1, 1) + clean, 0, 1)   (dataset_denoising.py:140-144,151)."""
    return (noise.clamp(-1.0, 1.0) + clean).clamp(0.0, 1.0)


def psnr(estimate: torch.Tensor, target: torch.Tensor, data_range: float = 1.0) -> float:
    """10 log10(R^2 / MSE) after clamping the estimate to [0, 1] (test_denoising.py:220-226,334-343)."""
    mse = torch.mean((estimate.clamp(0.0, 1.0).double() - target.double()) ** 2).item()
    return float("inf") if mse == 0 else 10.0 * float(np.log10(data_range ** 2 / mse))
