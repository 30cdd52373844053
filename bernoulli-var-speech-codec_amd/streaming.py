"""Stateful chunked (streaming) encode / decode on top of the same kernels (SURVEY.md 8f rank 1,
BASELINE.json configs[4]).

The reference facade has no streaming mode: it zero-initialises the GRU state on every call
(bvrnn_codec_model.py:60,69) and never returns the state after the last frame (bvrnn.py:205).  The
primitives are causal, though, so chunked processing reproduces the offline result (code bits identical; waveform
to rounding, <= 1e-6 - long offline decodes batch two dot-product halves over all frames, short hops do not):

* front-end: frame t reads samples [256t-256, 256t+768) -> it is emitted as soon as those samples
  have arrived (algorithmic look-ahead 768 samples = 34.8 ms, README.md:19); the last frames of an
  utterance, which need the right reflect padding, are emitted by ``flush()``;
* BVRNN: the GRU state is carried from chunk to chunk (``bvc_bvrnn_encode/decode`` take h0, return hT);
* vocoder: the generator is causal, so the library keeps the last 64 rows of every activation tensor
  (``bvc_vocoder_stream_*``, include/bvcodec.h) and a hop computes only the rows of its new frames -
  256 samples per frame, bit-for-bit the kernels of the offline path.  ``incremental=False`` selects
  the older, stateless scheme instead: every output sample depends on at most ``CONTEXT_FRAMES`` past
  mel frames (conv_pre 6 frames + per stage 12*(k-1) = 120 samples of AMP halo + 1 sample of
  transposed conv), so each hop re-runs the generator over [context | new frames] and keeps the new
  samples (≈4x the work per 20 ms hop).  The tail beyond the last full frame (``flush``) always uses
  the context scheme.

``tests/test_gpu_streaming.py`` checks that any chunking reproduces the offline encode()/decode().
"""
import ctypes
import math

import torch

from . import _abi
from .model import SCALING

CONTEXT_FRAMES = 26      # ceil(6 + 1 + 15 + 1/8 + 120/64 + 1/64 + 120/128 + 1/128 + 120/256 + 6/256)


class StreamingEncoder:
    def __init__(self, model, batch, bitrate, device=None):
        self.m = model
        self.B = batch
        self.bits = model.bits_per_frame(bitrate)
        eng = model.engine(None if device is None else torch.empty(0, device=device))
        self.dev = eng.device
        c = model.conf
        self.hop, self.win, self.pl = c["hopsize"], c["winsize"], c["mel_pad_left"]
        self.h = torch.zeros(1, batch, c["h_dim"], device=self.dev)
        self.buf = torch.empty(batch, 0, device=self.dev)      # samples from index self.s0 on
        self.s0 = 0                                            # global index of buf[:, 0] (multiple of hop)
        self.n = 0                                             # samples received
        self.frames = 0                                        # frames emitted

    def _emit(self, upto, total_len=None):
        """Encode frames [self.frames, upto); total_len: utterance length when flushing."""
        k = upto - self.frames
        if k <= 0:
            return torch.empty(self.B, 0, self.m.conf["z_dim"], device=self.dev)
        mel = self.m.mel_spectrogram(self.buf)                 # local frames; reflect pads only matter at the ends
        f0 = self.frames - self.s0 // self.hop
        mel = mel[:, f0:f0 + k].contiguous()
        bits = torch.full((self.B, k), self.bits, device=self.dev)
        codes, self.h = self.m.bvrnn.encode_stateful(mel, bits, self.h)
        self.frames = upto
        # keep what later frames still read: from sample hop*frames - 2*hop (so the next frame is local frame 2)
        keep_from = max(0, self.hop * self.frames - 2 * self.hop)
        if keep_from > self.s0:
            self.buf = self.buf[:, keep_from - self.s0:].contiguous()
            self.s0 = keep_from
        return codes

    @torch.no_grad()
    def push(self, x):
        """x (B, n) new samples -> codes (B, k, z_dim) of the frames completed by them (k may be 0)."""
        x = x.to(self.dev, torch.float32)
        self.buf = torch.cat([self.buf, x], 1)
        self.n += x.shape[1]
        pr = self.win - self.pl - self.hop                     # right look-ahead beyond the hop (512)
        complete = 0 if self.n < self.hop + pr else (self.n - self.hop - pr) // self.hop + 1
        if self.buf.shape[1] <= 2 * self.hop:                  # the front-end kernel needs L > 512
            complete = self.frames
        return self._emit(complete)

    @torch.no_grad()
    def flush(self):
        """End of utterance: the remaining floor(L/hop) - emitted frames (right reflect padding)."""
        total = self.n // self.hop
        if self.buf.shape[1] <= 2 * self.hop and total > self.frames:
            raise RuntimeError("utterance too short for the reflect padding of the STFT front-end")
        return self._emit(total)


class VocoderStream:
    """Handle of the library's incremental generator state for `batch` parallel streams."""

    def __init__(self, engine, batch, max_frames_per_push):
        self.eng = engine
        self.B = batch
        self.kmax = max_frames_per_push
        self.spf = math.prod(engine.conf["vocoder_config"]["upsample_rates"])
        h = ctypes.c_void_p()
        with torch.cuda.device(engine.device):
            _abi.check(engine.lib.bvc_vocoder_stream_create(engine.handle, batch, max_frames_per_push, ctypes.byref(h)))
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.eng.lib.bvc_vocoder_stream_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def reset(self):
        with torch.cuda.device(self.eng.device):
            _abi.check(self.eng.lib.bvc_vocoder_stream_reset(self.handle, self.eng.stream()))

    def push(self, mel, scale_div=1.0):
        """mel (B, k, num_mels) time-major on the engine's device -> wav (B, k*256)."""
        B, k, _ = mel.shape
        assert B == self.B
        out = torch.empty(B, k * self.spf, device=self.eng.device)
        with torch.cuda.device(self.eng.device):
            for f in range(0, k, self.kmax):                   # longer chunks go through in kmax-frame pieces
                n = min(self.kmax, k - f)
                piece = mel[:, f:f + n].contiguous()
                dst = out if n == k else torch.empty(B, n * self.spf, device=self.eng.device)
                _abi.check(self.eng.lib.bvc_vocoder_stream_push(self.handle, _abi.ptr(piece), n, float(scale_div),
                                                                _abi.ptr(dst), self.eng.stream()))
                if dst is not out:
                    out[:, f * self.spf:(f + n) * self.spf] = dst
        return out


class StreamingDecoder:
    def __init__(self, model, batch, device=None, incremental=True, max_frames_per_push=8):
        self.m = model
        self.B = batch
        eng = model.engine(None if device is None else torch.empty(0, device=device))
        self.dev = eng.device
        c = model.conf
        self.h = torch.zeros(1, batch, c["h_dim"], device=self.dev)
        self.ctx = torch.empty(batch, 0, c["num_mels"], device=self.dev)     # last CONTEXT_FRAMES mel frames
        self.spf = math.prod(c["vocoder_config"]["upsample_rates"])         # samples per frame (256)
        self.tail = None
        self.voc = VocoderStream(eng, batch, max_frames_per_push) if incremental else None

    @torch.no_grad()
    def push(self, codes):
        """codes (B, k, z_dim) -> wav (B, 256*k): the samples of exactly those frames."""
        k = codes.shape[1]
        if k == 0:
            return torch.empty(self.B, 0, device=self.dev)
        mel, self.h = self.m.bvrnn.decode(codes.to(self.dev, torch.float32), self.h)
        allmel = torch.cat([self.ctx, mel], 1)
        nctx = self.ctx.shape[1]
        self.ctx = allmel[:, -CONTEXT_FRAMES:].contiguous()
        if self.voc is not None:
            self.tail = None
            return self.voc.push(mel, SCALING)
        wav = self.m.vocoder(allmel, 10 ** 12, _scale_div=SCALING, _time_major=True)[:, 0]
        out = wav[:, self.spf * nctx: self.spf * (nctx + k)]
        self.tail = wav[:, self.spf * (nctx + k):]             # partial sums beyond the last frame (models.py:238)
        return out

    @torch.no_grad()
    def flush(self, n_extra):
        """Up to 294 samples beyond the last full frame (what decode(codes, length) returns past 256*T)."""
        if n_extra <= 0 or self.ctx.shape[1] == 0:
            return torch.empty(self.B, 0, device=self.dev)
        if self.tail is None:                                  # incremental mode: the tail comes from the context
            wav = self.m.vocoder(self.ctx, 10 ** 12, _scale_div=SCALING, _time_major=True)[:, 0]
            self.tail = wav[:, self.spf * self.ctx.shape[1]:]
        return self.tail[:, :max(0, n_extra)]


class _DeviceView:
    """Raw device memory owned by the library, exposed to torch through the CUDA array interface."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


class StreamingCodec:
    """BASELINE configs[4]: `batch` parallel streams, a fixed hop of new samples per tick, encode + decode of the frames
    each hop completes in ONE library call (``bvc_stream_codec_tick``: one persistent launch per recurrence where that
    kernel is available, else a hipGraph of launch-per-layer kernels replayed once the streams are warm).  State (sample buffer, both GRU states, the generator's activation history) lives in the library.

    ``push(x)`` with x (batch, hop) returns (codes (batch, k, z_dim), wav (batch, 256 k)) for the k frames completed;
    they equal the offline ``encode`` / ``decode`` of the whole signal on those frames (tests/test_gpu_streaming.py).
    The returned tensors are views of the state's output buffers: valid until the next push."""

    def __init__(self, model, batch, bitrate, hop=441, device=None):
        eng = model.engine(None if device is None else torch.empty(0, device=device))
        self.eng, self.B, self.hop = eng, batch, hop
        self.dev = eng.device
        self.z = model.conf["z_dim"]
        self.spf = math.prod(model.conf["vocoder_config"]["upsample_rates"])
        self.stream = torch.cuda.Stream(self.dev)          # a tick may be captured into a hipGraph: not possible on the default stream
        h = ctypes.c_void_p()
        with torch.cuda.device(self.dev):
            _abi.check(eng.lib.bvc_stream_codec_create(eng.handle, batch, hop, float(model.bits_per_frame(bitrate)), float(SCALING),
                                                       float(SCALING), ctypes.byref(h)))
        self.handle = h
        pin, pc, pw, kmax = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int32()
        _abi.check(eng.lib.bvc_stream_codec_buffers(h, ctypes.byref(pin), ctypes.byref(pc), ctypes.byref(pw), ctypes.byref(kmax)))
        self.kmax = kmax.value
        self._in = torch.as_tensor(_DeviceView(pin.value, (batch, hop)), device=self.dev)
        self._codes_ptr, self._wav_ptr = pc.value, pw.value
        self.frames = 0

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.eng.lib.bvc_stream_codec_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    @torch.no_grad()
    def push(self, x):
        assert tuple(x.shape) == (self.B, self.hop)
        cur = torch.cuda.current_stream(self.dev)
        self.stream.wait_stream(cur)
        k = ctypes.c_int32()
        with torch.cuda.stream(self.stream), torch.cuda.device(self.dev):
            self._in.copy_(x.to(self.dev, torch.float32), non_blocking=True)
            _abi.check(self.eng.lib.bvc_stream_codec_tick(self.handle, ctypes.byref(k), ctypes.c_void_p(self.stream.cuda_stream)))
        cur.wait_stream(self.stream)
        k = k.value
        self.frames += k
        if k == 0:
            return torch.empty(self.B, 0, self.z, device=self.dev), torch.empty(self.B, 0, device=self.dev)
        codes = torch.as_tensor(_DeviceView(self._codes_ptr, (self.B, k, self.z)), device=self.dev)
        wav = torch.as_tensor(_DeviceView(self._wav_ptr, (self.B, k * self.spf)), device=self.dev)
        return codes, wav
