#!/usr/bin/env python3
"""Writes tests/golden/g2_melbank_independent.npz: the Slaney mel filterbank computed by an
implementation that is independent of this repository - transformers.audio_utils.mel_filter_bank -
so that the restated librosa.filters.mel (oracle/melbank.py, bvcodec/melbank.py) is pinned without
importing transformers (a multi-minute import on a cold cache) in every test run."""
import os

import numpy as np
from transformers.audio_utils import mel_filter_bank

m = mel_filter_bank(513, 80, 0.0, 8000.0, 22050, norm="slaney", mel_scale="slaney").T
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "g2_melbank_independent.npz"),
                    mel_basis=np.asarray(m, dtype=np.float64))
print(m.shape, np.count_nonzero(m))
