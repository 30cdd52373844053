"""TOML configuration of the codec path.

Reads the same file format and key names the reference facade reads at
bvrnn_codec_model.py:27-36,49-59 (configs/config_varBitRate.toml / config_64bit.toml load
unchanged); training-only keys are ignored.  ``AttrDict`` mirrors
third_party/BigVGAN/env.py:8-11 (attribute-style access to ``vocoder_config``).
"""
import os

try:                        # Python >= 3.11
    import tomllib as _toml
except ModuleNotFoundError:  # this image: Python 3.10 + tomli
    import tomli as _toml

_HERE = os.path.abspath(os.path.dirname(__file__))
DEFAULT_CONFIG = os.path.join(_HERE, "configs", "codec_varbitrate.toml")
DEFAULT_CONFIG_64BIT = os.path.join(_HERE, "configs", "codec_64bit.toml")

_REQUIRED = ("var_bit", "fs", "winsize", "hopsize", "num_mels", "fmin", "fmax", "mel_pad_left",
             "h_dim", "z_dim", "vocoder_config")
_REQUIRED_VOC = ("num_mels", "upsample_rates", "upsample_kernel_sizes", "upsample_initial_channel",
                 "resblock_kernel_sizes", "resblock_dilation_sizes")


class AttrDict(dict):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self


def load_config(path):
    with open(path, "rb") as f:
        conf = _toml.load(f)
    for k in _REQUIRED:
        if k not in conf:
            raise KeyError(f"config {path}: missing key '{k}'")
    for k in _REQUIRED_VOC:
        if k not in conf["vocoder_config"]:
            raise KeyError(f"config {path}: missing key 'vocoder_config.{k}'")
    check_supported(conf)
    return conf


def check_supported(conf):
    """The HIP path covers what the two shipped TOMLs select; anything else fails loudly."""
    v = conf["vocoder_config"]
    bad = []
    if v.get("resblock", "1") != "1":
        bad.append("vocoder_config.resblock must be '1'")
    if v.get("activation", "snakebeta") != "snakebeta" or not v.get("snake_logscale", True):
        bad.append("only activation='snakebeta' with snake_logscale=true is implemented")
    if any(v.get("layers_sym", [False])) or v.get("pre_sym", False) or v.get("post_sym", False):
        bad.append("only causal (non-symmetric) layers are implemented")
    if any(v.get("layers_antialias", [False])) or v.get("antialias_post", False):
        bad.append("anti-aliased activations are not implemented (both shipped configs disable them)")
    for u, k in zip(v["upsample_rates"], v["upsample_kernel_sizes"]):
        if k != 2 * u:
            bad.append(f"transposed conv kernel {k} must be 2 x stride {u}")
    if conf["winsize"] != 1024 or conf["hopsize"] != 256:
        bad.append("front-end kernel is specialised for winsize=1024, hopsize=256")
    if conf["mel_pad_left"] < 0 or conf["mel_pad_left"] > conf["winsize"] - conf["hopsize"]:
        bad.append("mel_pad_left out of range")
    if conf["z_dim"] % 16 or conf["h_dim"] % 16 or conf["num_mels"] % 16:
        bad.append("z_dim, h_dim and num_mels must be multiples of 16")
    if bad:
        raise ValueError("unsupported configuration: " + "; ".join(bad))
