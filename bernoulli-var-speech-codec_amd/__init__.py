"""MI355X-native BVRNN speech-codec hot path (STFT/log-mel -> BVRNN coder -> causal BigVGAN).

Drop-in for the reference facade ``bvrnn_codec_model.BVRNNCodecModel`` (bvrnn_codec_model.py:19-76):
same constructor, ``encode(x, bitrate)``, ``decode(codes, length)``, ``forward(x, bitrate)``.
All compute runs in hand-written gfx950 HIP kernels behind the C ABI of ``include/bvcodec.h``.
Imported as ``bvcodec`` (see ``bvcodec/__init__.py`` at the repository root).
"""
__all__ = ["BVRNNCodecModel", "load_config", "AttrDict"]


def __getattr__(name):              # lazy: config/synth stay importable without the HIP library
    if name == "BVRNNCodecModel":
        from .model import BVRNNCodecModel
        return BVRNNCodecModel
    if name in ("load_config", "AttrDict"):
        from . import config
        return getattr(config, name)
    raise AttributeError(name)
