"""Drop-in import shim: ``from bvrnn_codec_model import BVRNNCodecModel`` (the reference's module
name, bvrnn_codec_model.py:19) resolves to the MI355X implementation in ``bvcodec``."""
from bvcodec.model import SCALING, BVRNNCodecModel, default_chkpt_bvrnn, default_chkpt_vocoder, default_config  # noqa: F401
