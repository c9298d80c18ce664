"""Shared helpers for the parity tests."""
import numpy as np
import torch

from noisediff_amd import synth
from noisediff_amd.spec import attention_param_spec, noisediff_param_spec


def state_dict(dim, seed=0, mid_attn=False):
    sd = synth.make_state_dict(noisediff_param_spec(dim), seed)
    if mid_attn:
        sd.update(synth.make_state_dict(attention_param_spec("mid_attn", 8 * dim), seed))
    return sd


def noise_fn(seed, batch, channels, size, first_sample=0):
    def fn(i, shape):
        return synth.make_noise(seed, f"noise.{i}", batch, channels, size, first_sample)
    return fn


def sub(t, n=4096):
    f = torch.as_tensor(t).reshape(-1)
    step = max(f.numel() // n, 1)
    return f[::step][:n].numpy()


def rel_err(a, b):
    """max |a-b| / max(1, max|b|): 'relative fp32 tolerance' on O(1) tensors."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(1.0, float(np.max(np.abs(b)))))
