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


ELEM_RTOL = 1e-3          # the north-star's "1e-3 relative fp32 tolerance", element by element ...
ELEM_ATOL = 1e-5          # ... with this floor where the reference crosses zero: |a - b| <= ELEM_ATOL + ELEM_RTOL * |b| for EVERY element


def elem_excess(a, b, rtol=ELEM_RTOL):
    """max over the elements of |a - b| - rtol * |b|: the absolute floor the element-wise criterion |a - b| <= atol + rtol * |b| would need to hold everywhere."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) - rtol * np.abs(b)))


def close(a, b, tol, atol=ELEM_ATOL, rtol=ELEM_RTOL, log_only=False):
    """Both parity criteria of DESIGN section 6: rel_err(a, b) < tol (max |a - b| / max(1, max |b|)) AND zero elements outside |a - b| <= atol + rtol * |b|.
    Set ND_TEST_ELEM_LOG=<file> to log the measured figures of every call.  ``log_only=True`` (an explicit argument of the caller, never the environment:
    ADVICE r5) records the element-wise figure without asserting it -- for the tools that measure the floors."""
    import os
    r, x = rel_err(a, b), elem_excess(a, b, rtol)
    if os.environ.get("ND_TEST_ELEM_LOG"):
        with open(os.environ["ND_TEST_ELEM_LOG"], "a") as f:
            f.write(f"{os.environ.get('PYTEST_CURRENT_TEST', '?')}: rel_err {r:.3e} (tol {tol:g}), element-wise floor needed {x:.3e} (atol {atol:g})\n")
    assert r < tol, f"rel_err {r:.3e} >= {tol:g}"
    assert x <= atol or log_only, f"element-wise: |a - b| exceeds {rtol:g} * |b| by {x:.3e} > atol {atol:g}"
    return True
