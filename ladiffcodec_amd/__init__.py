"""MI355X-native LaDiffCodec decode path (HIP/CDNA4 kernels behind a C ABI).

Importing the package is cheap and does not need a GPU; the HIP library is loaded on first use by
`ladiffcodec_amd.lib.load()` and that call fails loudly when `libladiffcodec.so` has not been built
(there is no CPU fallback in this package).
"""
from .spec import CodecConfig, UnetConfig  # noqa: F401

__all__ = ["CodecConfig", "UnetConfig"]
