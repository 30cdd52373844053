"""Importable alias of the ``bernoulli-var-speech-codec_amd`` package.

The product directory carries the reference's repository name (hyphens), which Python cannot
import by name; this stub points its submodule search path at that directory, so
``import bvcodec`` / ``from bvcodec import BVRNNCodecModel`` work from the repository root.
"""
import os as _os

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "bernoulli-var-speech-codec_amd")
__path__.insert(0, _REAL)

with open(_os.path.join(_REAL, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_REAL, "__init__.py"), "exec"))
