from .modules import Generator, ResnetBlock, WNConv1d, WNConvTranspose1d, weights_init

__all__ = ["Generator", "ResnetBlock", "WNConv1d", "WNConvTranspose1d", "weights_init"]
