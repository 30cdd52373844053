"""Oracle: STFT + log-mel front-end (test infrastructure, see oracle/__init__.py).

Follows ``mel_spectrogram`` third_party/BigVGAN/meldataset.py:60-95 with the arguments the
codec facade passes at bvrnn_codec_model.py:49-56 (center=False, padding_left=256), and
``dynamic_range_compression_torch`` meldataset.py:38-39.
"""
import torch

from .melbank import mel_filterbank

SCALING = 10 ** (-10 / 20)          # bvrnn_codec_model.py:17


def log_mel(x, n_fft=1024, num_mels=80, sampling_rate=22050, hop_size=256, win_size=1024,
            fmin=0, fmax=8000, padding_left=256, dtype=torch.float32, mel_basis=None):
    """x: (B, L) waveform ALREADY scaled by the caller.  Returns (B, num_mels, T)."""
    x = torch.as_tensor(x).to(dtype)
    if mel_basis is None:
        mel_basis = torch.from_numpy(mel_filterbank(sampling_rate, n_fft, num_mels, fmin, fmax))
    mel_basis = mel_basis.to(dtype)                                   # meldataset.py:69 (.float())
    window = torch.hann_window(win_size, dtype=torch.float32).to(dtype)   # meldataset.py:70
    if padding_left == -1:                                            # meldataset.py:72-75
        pl = pr = (n_fft - hop_size) // 2
    else:                                                             # meldataset.py:76-78
        pl = padding_left
        pr = win_size - padding_left - hop_size
    y = torch.nn.functional.pad(x.unsqueeze(1), (pl, pr), mode="reflect").squeeze(1)   # :80-81
    stft = torch.stft(y, n_fft, hop_length=hop_size, win_length=win_size, window=window,
                      center=False, pad_mode="reflect", normalized=False, onesided=True,
                      return_complex=True)                            # :84-85
    spec = torch.view_as_real(stft)
    spec = torch.sqrt(spec.pow(2).sum(-1) + 1e-9)                     # :86-87
    spec = torch.matmul(mel_basis, spec)                              # :89
    return torch.log(torch.clamp(spec, min=1e-5))                     # :38-39, :90
