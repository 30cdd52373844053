"""Slaney-scale triangular mel filterbank (the matrix ``librosa.filters.mel`` returns for
sr=22050, n_fft=1024, n_mels=80, fmin=0, fmax=8000, htk=False, norm='slaney'; call site
third_party/BigVGAN/meldataset.py:68).  librosa is not a dependency of this package: the published
construction is implemented here in float64 and cast to float32, then handed to the HIP library,
which stores it sparse (727 non-zero weights, bins 0..371)."""
import numpy as np

_F_SP = 200.0 / 3.0          # Hz per mel below the 1 kHz knee
_KNEE_HZ = 1000.0
_KNEE_MEL = _KNEE_HZ / _F_SP
_LOGSTEP = np.log(6.4) / 27.0


def _hz_to_mel(hz):
    hz = np.atleast_1d(np.asarray(hz, dtype=np.float64))
    out = hz / _F_SP
    hi = hz >= _KNEE_HZ
    out[hi] = _KNEE_MEL + np.log(hz[hi] / _KNEE_HZ) / _LOGSTEP
    return out


def _mel_to_hz(mel):
    mel = np.atleast_1d(np.asarray(mel, dtype=np.float64))
    out = mel * _F_SP
    hi = mel >= _KNEE_MEL
    out[hi] = _KNEE_HZ * np.exp(_LOGSTEP * (mel[hi] - _KNEE_MEL))
    return out


def slaney_mel_basis(sample_rate, n_fft, n_mels, fmin, fmax):
    """-> float32 (n_mels, n_fft//2 + 1)."""
    if fmax is None:
        fmax = sample_rate / 2.0
    n_bins = n_fft // 2 + 1
    bin_hz = np.linspace(0.0, sample_rate / 2.0, n_bins)
    lo, hi = _hz_to_mel(fmin)[0], _hz_to_mel(fmax)[0]
    edges = _mel_to_hz(np.linspace(lo, hi, n_mels + 2))            # n_mels + 2 band edges in Hz
    left, centre, right = edges[:-2, None], edges[1:-1, None], edges[2:, None]
    rising = (bin_hz[None, :] - left) / (centre - left)
    falling = (right - bin_hz[None, :]) / (right - centre)
    tri = np.clip(np.minimum(rising, falling), 0.0, None)
    tri *= 2.0 / (right - left)                                    # Slaney area normalisation
    return tri.astype(np.float32)
