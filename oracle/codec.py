"""Oracle: the codec facade (test infrastructure, see oracle/__init__.py).

Follows ``BVRNNCodecModel.encode/decode/forward`` bvrnn_codec_model.py:44-76: scale by
-10 dB (:17,:49), log-mel (:49-56), bits per frame = np.round(bitrate*hop/fs) (:58-59),
zero initial state (:60,:69), BVRNN encode/decode (:61,:70), vocoder and un-scale (:71).
"""
import numpy as np
import torch

from . import bigvgan, bvrnn, frontend

SCALING = frontend.SCALING


class OracleCodec:
    """conf: parsed TOML dict; vrnn_sd / gen_sd: the reference state dicts."""

    def __init__(self, conf, vrnn_sd, gen_sd, dtype=torch.float32):
        self.conf, self.vrnn, self.gen, self.dtype = conf, vrnn_sd, gen_sd, dtype

    def mel(self, x):
        c = self.conf
        x = torch.as_tensor(x).to(self.dtype)
        m = frontend.log_mel(x * SCALING, n_fft=c["winsize"], num_mels=c["num_mels"],
                             sampling_rate=c["fs"], hop_size=c["hopsize"], win_size=c["winsize"],
                             fmin=c["fmin"], fmax=c["fmax"], padding_left=c["mel_pad_left"],
                             dtype=self.dtype)
        return m.permute(0, 2, 1)                                    # (B, T, 80)

    def bits_per_frame(self, bitrate):
        return float(np.round(bitrate * self.conf["hopsize"] / self.conf["fs"]))

    def encode(self, x, bitrate, full=False):
        xmel = self.mel(x)
        B, T, _ = xmel.shape
        bits = self.bits_per_frame(bitrate) * torch.ones((B, T))
        h0 = torch.zeros(B, self.conf["h_dim"])
        r = bvrnn.encode(self.vrnn, xmel, bits, h0, var_bit=self.conf["var_bit"], dtype=self.dtype)
        return r if full else r["codes"]

    def decode_mel(self, codes):
        h0 = torch.zeros(codes.shape[0], self.conf["h_dim"])
        return bvrnn.decode(self.vrnn, codes, h0, dtype=self.dtype)["mel"]

    def decode(self, codes, length):
        xmel = self.decode_mel(torch.as_tensor(codes))
        wav = bigvgan.forward(self.gen, self.conf["vocoder_config"], xmel.permute(0, 2, 1), length,
                              dtype=self.dtype)
        return wav.squeeze(1) / SCALING

    def forward(self, x, bitrate):
        x = torch.as_tensor(x)
        return self.decode(self.encode(x, bitrate), x.shape[1])
