"""MI355X-native sampling hot path of IVRL/NoiseDiff."""
__version__ = "0.1.0"

__all__ = ["GaussianDiffusion", "NoiseDiffNet", "UNet_PosEmbV2", "UNet_PosEmbV2_NoPosition", "UNet_PosEmbV2_CameraCond", "LSID",
           "TrainableNoiseDiffNet", "__version__"]


def __getattr__(name):
    # lazy: importing the package must not require torch.cuda or the built library
    if name == "GaussianDiffusion":
        from .diffusion import GaussianDiffusion
        return GaussianDiffusion
    if name in ("NoiseDiffNet", "UNet_PosEmbV2", "UNet_PosEmbV2_NoPosition", "UNet_PosEmbV2_CameraCond"):
        from . import net
        return getattr(net, name)
    if name == "LSID":
        from .lsid import LSID
        return LSID
    if name == "TrainableNoiseDiffNet":
        from .trainable import TrainableNoiseDiffNet
        return TrainableNoiseDiffNet
    raise AttributeError(name)
