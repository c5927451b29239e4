"""stt_amd -- MI355X-native engine for the coqui-ai/STT streaming inference hot path.

The product is the C-ABI shared library stt_amd/lib/libstt.so (HIP kernels for gfx950 behind
include/coqui-stt.h); this package is only its Python face.  Build: `python -m stt_amd.build`.
"""
from .model import Decoder, Model, Stream  # noqa: F401

__all__ = ["Model", "Stream", "Decoder"]
