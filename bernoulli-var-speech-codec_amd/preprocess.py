"""GPU pre-processing of the reference's example script (example.py:12-17): first channel, rational-rate
polyphase resampling to 22.05 kHz (``scipy.signal.resample_poly(speech, 22050, fs)``) and peak
normalisation (``speech / np.max(np.abs(speech))``).  SURVEY.md 8(f) rank 3.

The low-pass design restates scipy's defaults (``firwin(2*10*max(up,down)+1, 1/max(up,down),
window=('kaiser', 5.0))``, unit DC gain, scaled by ``up``, zero-padded so the output is centred) in numpy
float64; the filtering itself (upfirdn + slicing) runs in ``bvc_resample_poly`` on the GPU.
"""
import ctypes
import math

import numpy as np
import torch

from . import _abi


def kaiser_lowpass(numtaps, cutoff, beta=5.0):
    """scipy.signal.firwin(numtaps, cutoff, window=('kaiser', beta)) for a low-pass (cutoff rel. to Nyquist)."""
    m = np.arange(numtaps, dtype=np.float64) - (numtaps - 1) / 2.0
    h = cutoff * np.sinc(cutoff * m)
    h *= np.kaiser(numtaps, beta)
    return h / h.sum()                       # unit gain at DC


def design(up, down):
    """-> (up, down, h_padded float64, n_pre_remove) exactly as scipy.signal.resample_poly arranges them."""
    g = math.gcd(int(up), int(down))
    up, down = int(up) // g, int(down) // g
    max_rate = max(up, down)
    half_len = 10 * max_rate
    h = kaiser_lowpass(2 * half_len + 1, 1.0 / max_rate) * up
    n_pre_pad = down - half_len % down
    n_pre_remove = (half_len + n_pre_pad) // down
    return up, down, np.concatenate([np.zeros(n_pre_pad), h]), n_pre_remove


def resample_poly(x, up, down):
    """x (B, L) float tensor on the GPU -> (B, ceil(L*up/down)) float32."""
    if x.device.type != "cuda":
        raise RuntimeError("bvcodec.preprocess runs on the GPU only")
    up, down, h, n_pre_remove = design(up, down)
    x = x.detach().to(torch.float32).contiguous()
    if up == down == 1:
        return x.clone()
    B, L = x.shape
    n_out = (L * up + down - 1) // down
    # the filter tail must cover every kept output (scipy's n_post_pad loop)
    need = (n_out + n_pre_remove - 1) * down - (L - 1) * up + 1
    if need > len(h):
        h = np.concatenate([h, np.zeros(need - len(h))])
    hd = torch.from_numpy(h).to(x.device)
    y = torch.empty(B, n_out, device=x.device)
    lib = _abi.load()
    with torch.cuda.device(x.device):
        _abi.check(lib.bvc_resample_poly(_abi.ptr(x), B, L, ctypes.c_void_p(hd.data_ptr()), len(h), up, down,
                                         n_pre_remove, _abi.ptr(y), n_out, _abi.current_stream(x.device)))
    return y


def peak_normalize(x):
    """x (B, L) -> x / max|x| per utterance (new tensor)."""
    y = x.detach().to(torch.float32).contiguous().clone()
    lib = _abi.load()
    with torch.cuda.device(y.device):
        _abi.check(lib.bvc_peak_normalize(_abi.ptr(y), y.shape[0], y.shape[1], _abi.current_stream(y.device)))
    return y


def prepare_speech(x, fs_in, fs_out=22050):
    """example.py:12-17 for a batch: resample fs_in -> fs_out and peak-normalise."""
    return peak_normalize(resample_poly(x, fs_out, fs_in))
