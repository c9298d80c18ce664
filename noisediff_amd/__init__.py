"""MI355X-native sampling hot path of IVRL/NoiseDiff."""
__version__ = "0.1.0"

__all__ = ["GaussianDiffusion", "NoiseDiffNet", "LSID", "__version__"]


def __getattr__(name):
    # lazy: importing the package must not require torch.cuda or the built library
    if name == "GaussianDiffusion":
        from .diffusion import GaussianDiffusion
        return GaussianDiffusion
    if name == "NoiseDiffNet":
        from .net import NoiseDiffNet
        return NoiseDiffNet
    if name == "LSID":
        from .lsid import LSID
        return LSID
    raise AttributeError(name)
