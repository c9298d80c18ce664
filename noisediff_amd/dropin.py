"""Run the reference's own scripts on the HIP sampler with ZERO edits to the reference tree.

    python -m noisediff_amd.dropin /path/to/NoiseDiff/test_diffusion.py --with_camera_settings --beta_schedule sigmoid2 ...

The reference binds its two plug-in interfaces by *import name* (SURVEY 8b):
  * ``from models.denoising_diffusion_pytorch import GaussianDiffusion``     (models/trainer_diffusion.py:28)
  * ``getattr(models.archs.<x>_arch, args.net_name)`` over every ``*_arch.py`` file (models/modules.py:20-41)
``install()`` puts one finder in front of ``sys.meta_path`` that
  * answers ``models.denoising_diffusion_pytorch`` with a module exporting this package's ``GaussianDiffusion``
    (the reference file is not executed, so its unused torchvision / ema_pytorch imports are not needed for it), and
  * lets ``models.archs.Diffusion_arch`` / ``others_arch`` load normally, then rebinds the class names this package
    implements (``NoiseDiffNet``, ``UNet_PosEmbV2*``) to a constructor that returns the HIP network for inference
    (``args.phase != 'train'``) and the reference's own differentiable class for training -- its 3x3 convolutions routed
    through the HIP library forward and backward (noisediff_amd/train.py; ND_TRAIN_ACCEL=0 keeps them on PyTorch) -- whose
    checkpoints load into the HIP network unchanged (same state dict).
``main()`` then runs the script with ``runpy`` as ``__main__`` (same ``sys.argv``, script directory first on ``sys.path``),
which is what ``python test_diffusion.py ...`` does.  Nothing is written to the reference tree.
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.util
import os
import runpy
import sys
import types

DIFFUSION_MODULE = "models.denoising_diffusion_pytorch"
ARCH_MODULES = ("models.archs.Diffusion_arch", "models.archs.others_arch")
HIP_CLASSES = ("NoiseDiffNet", "UNet_PosEmbV2", "UNet_PosEmbV2_NoPosition", "UNet_PosEmbV2_CameraCond")


def _hip_or_reference(name: str, reference_cls):
    """Constructor with the registry's calling convention ``cls(args)`` (models/modules.py:41)."""
    def construct(args):
        if getattr(args, "phase", "test") == "train":
            net = reference_cls(args)                # autograd path: the reference's own nn.Module ...
            if os.environ.get("ND_TRAIN_ACCEL", "1") != "0":
                from noisediff_amd import train
                train.accelerate(net)                # ... with its 3x3 convolutions (85 % of the FLOPs) forward and backward on the HIP library
            return net
        import noisediff_amd
        return getattr(noisediff_amd, name)(args)
    construct.__name__ = construct.__qualname__ = name
    construct.reference_class = reference_cls
    construct.__doc__ = f"{name}(args): noisediff_amd.{name} (HIP) unless args.phase == 'train' (then the reference class)."
    return construct


class _DiffusionLoader(importlib.abc.Loader):
    def create_module(self, spec):
        return None

    def exec_module(self, module):
        from noisediff_amd import diffusion
        for k in ("GaussianDiffusion", "make_betas", "make_buffers", "BUFFER_NAMES"):
            setattr(module, k, getattr(diffusion, k))
        module.__doc__ = "noisediff_amd overlay of models/denoising_diffusion_pytorch.py (see noisediff_amd/dropin.py)"


class _ArchLoader(importlib.abc.Loader):
    def __init__(self, inner):
        self.inner = inner

    def create_module(self, spec):
        return self.inner.create_module(spec)

    def exec_module(self, module):
        self.inner.exec_module(module)
        for name in HIP_CLASSES:
            ref = getattr(module, name, None)
            if isinstance(ref, type):
                setattr(module, name, _hip_or_reference(name, ref))


class OverlayFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname == DIFFUSION_MODULE:
            return importlib.util.spec_from_loader(fullname, _DiffusionLoader(), origin="noisediff_amd.dropin")
        if fullname in ARCH_MODULES:
            for finder in sys.meta_path:
                if finder is self or not hasattr(finder, "find_spec"):
                    continue
                spec = finder.find_spec(fullname, path, target)
                if spec is not None and spec.loader is not None:
                    spec.loader = _ArchLoader(spec.loader)
                    return spec
        return None


def install() -> OverlayFinder:
    """Idempotent: put the overlay finder first on sys.meta_path."""
    for f in sys.meta_path:
        if isinstance(f, OverlayFinder):
            return f
    f = OverlayFinder()
    sys.meta_path.insert(0, f)
    for name in (DIFFUSION_MODULE,) + ARCH_MODULES:
        if name in sys.modules:
            raise RuntimeError(f"{name} was imported before noisediff_amd.dropin.install(); install the overlay first")
    return f


def main(argv=None) -> None:
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        raise SystemExit(0 if argv else 2)
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script):
        raise SystemExit(f"noisediff_amd.dropin: no such script: {script}")
    install()
    sys.argv = [script] + argv[1:]
    sys.path.insert(0, os.path.dirname(script))      # what `python script.py` does
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
