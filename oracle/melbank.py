"""Oracle: Slaney mel filterbank + periodic Hann window (test infrastructure, see __init__).

Restates ``librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)`` with its defaults
(``htk=False``, ``norm='slaney'``) as called at third_party/BigVGAN/meldataset.py:68,
and ``torch.hann_window(win_size)`` (periodic) at meldataset.py:70.
"""
import numpy as np


def hz_to_mel_slaney(f):
    """Slaney (Auditory Toolbox) mel scale: linear below 1 kHz, log above."""
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3.0
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    with np.errstate(divide="ignore", invalid="ignore"):
        log_part = min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep
    return np.where(f >= min_log_hz, log_part, mels)


def mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3.0
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(sr=22050, n_fft=1024, n_mels=80, fmin=0.0, fmax=8000.0):
    """(n_mels, n_fft//2+1) float32 filterbank, built in float64 like librosa."""
    n_bins = n_fft // 2 + 1
    fftfreqs = np.linspace(0.0, sr / 2.0, n_bins)                      # librosa 0.8.1 fft_frequencies
    mel_pts = np.linspace(hz_to_mel_slaney(fmin), hz_to_mel_slaney(fmax), n_mels + 2)
    hz_pts = mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(hz_pts)
    ramps = hz_pts[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, n_bins), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (hz_pts[2:n_mels + 2] - hz_pts[:n_mels])             # norm='slaney'
    weights *= enorm[:, None]
    return weights.astype(np.float32)


def hann_periodic(n=1024):
    """torch.hann_window(n) (periodic=True): 0.5 - 0.5 cos(2 pi k / n), float32."""
    import torch
    return torch.hann_window(n, dtype=torch.float32).numpy()
