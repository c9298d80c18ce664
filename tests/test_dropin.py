"""The zero-edit overlay (noisediff_amd/dropin.py): the reference's import names resolve to this package's classes.

Two checks, each in a fresh interpreter (the overlay patches the import system):
  * a miniature tree with the reference's LAYOUT (models/modules.py-style registry scanning archs/*_arch.py by suffix,
    a script importing `from models.denoising_diffusion_pytorch import GaussianDiffusion`) written by this test;
  * the real reference tree when it is mounted (build container only; skipped on the GPU box).
"""
import os
import subprocess
import sys
import textwrap

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def _run(code, cwd, *args):
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""), PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code), *args], cwd=cwd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_overlay_on_a_tree_with_the_reference_layout(tmp_path):
    root = tmp_path / "tree"
    (root / "models" / "archs").mkdir(parents=True)
    (root / "models" / "__init__.py").write_text("")
    (root / "models" / "archs" / "__init__.py").write_text("")
    (root / "models" / "archs" / "Diffusion_arch.py").write_text(
        "import torch.nn as nn\nclass NoiseDiffNet(nn.Module):\n    def __init__(self, args):\n        super().__init__()\n        self.tag = 'tree'\n"
        "class Unrelated(nn.Module):\n    pass\n")
    (root / "models" / "denoising_diffusion_pytorch.py").write_text("raise ImportError('the overlay must answer this name')\n")
    (root / "models" / "modules.py").write_text(textwrap.dedent("""
        import importlib, os
        folder = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'archs')
        mods = [importlib.import_module('models.archs.' + f[:-3]) for f in sorted(os.listdir(folder)) if f.endswith('_arch.py')]
        def define_network(args):
            for m in mods:
                c = getattr(m, args.net_name, None)
                if c is not None:
                    return c(args)
            raise ValueError(args.net_name)
    """))
    (root / "run_me.py").write_text(textwrap.dedent("""
        import sys
        from types import SimpleNamespace
        from models.modules import define_network
        from models.denoising_diffusion_pytorch import GaussianDiffusion
        a = SimpleNamespace(net_name='NoiseDiffNet', dim=16, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False, phase=sys.argv[1])
        net = define_network(a)
        gd = GaussianDiffusion(net, image_size=32, timesteps=10, beta_schedule='sigmoid2') if sys.argv[1] == 'test' else None
        print('RESULT', type(net).__module__, type(net).__name__, GaussianDiffusion.__module__, sys.argv[1:], gd is not None and gd.num_timesteps)
    """))
    out = _run("from noisediff_amd import dropin; dropin.main()", str(tmp_path), str(root / "run_me.py"), "test", "--flag")
    assert "RESULT noisediff_amd.net NoiseDiffNet noisediff_amd.diffusion ['test', '--flag'] 10" in out
    out = _run("from noisediff_amd import dropin; dropin.main()", str(tmp_path), str(root / "run_me.py"), "train")
    assert "RESULT models.archs.Diffusion_arch NoiseDiffNet noisediff_amd.diffusion ['train'] False" in out


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted (GPU box)")
def test_overlay_on_the_real_reference_registry():
    """models/modules.py (the registry itself, unedited) hands out the HIP classes; training keeps the reference's."""
    out = _run("""
        import sys, types
        sys.dont_write_bytecode = True
        from types import SimpleNamespace
        import torch
        from noisediff_amd import dropin, train
        dropin.install()
        sys.path.insert(0, %r)
        import models.modules as M
        from models.denoising_diffusion_pytorch import GaussianDiffusion
        for name in ('NoiseDiffNet', 'UNet_PosEmbV2', 'UNet_PosEmbV2_NoPosition', 'UNet_PosEmbV2_CameraCond'):
            a = SimpleNamespace(net_name=name, dim=16, cond_dim=4, inp_dim=4, self_condition=False, normalize_condition=False, phase='test')
            hip = M.define_network(a)
            a.phase = 'train'
            ref = M.define_network(a)
            assert [(k, tuple(v.shape)) for k, v in hip.state_dict().items()] == [(k, tuple(v.shape)) for k, v in ref.state_dict().items()], name
            hip.load_state_dict(ref.state_dict(), strict=True)        # a checkpoint trained on the reference class loads unchanged
            n3 = sum(1 for m in ref.modules() if isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3))
            acc = sum(1 for m in ref.modules() if getattr(getattr(m, 'forward', None), '__func__', None) is train._hip_conv_forward)
            assert acc == n3 > 20, (name, acc, n3)                    # training: every 3x3 conv of the reference class goes through the HIP library
            print('RESULT', name, type(hip).__module__, type(ref).__module__)
        import torch
        gd = GaussianDiffusion(torch.nn.DataParallel(ref), image_size=32, timesteps=20, beta_schedule='sigmoid2')
        print('RESULT gd', type(gd).__module__, len(list(gd.buffers())))
        a = SimpleNamespace(net_name='LSID', phase='test')
        assert type(M.define_network(a)).__module__ == 'models.archs.SID_arch'       # other archs are untouched
    """ % REF, REPO)
    assert "RESULT NoiseDiffNet noisediff_amd.net models.archs.Diffusion_arch" in out
    assert "RESULT UNet_PosEmbV2_CameraCond noisediff_amd.net models.archs.others_arch" in out
    assert "RESULT gd noisediff_amd.diffusion 13" in out
