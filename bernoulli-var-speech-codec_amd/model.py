"""Drop-in facade: ``BVRNNCodecModel(config_path, bvrnn_chkpt_path, vocoder_chkpt_path)`` with
``encode(x, bitrate)``, ``decode(codes, length)``, ``forward(x, bitrate)`` - same names, argument
meaning and error behaviour as the reference class (bvrnn_codec_model.py:19-76) - plus the two
sub-operators the reference exposes as attributes (``model.bvrnn.encode/decode`` bvrnn.py:163-229,
``model.vocoder(x, length)`` models.py:207-238) and ``mel_spectrogram`` (meldataset.py:60-95).

All arithmetic runs in the gfx950 HIP library behind include/bvcodec.h; PyTorch only provides device
memory and streams.  Inference only: outputs carry no autograd graph.  Tensors that live on another
device (e.g. the CPU tensors of the reference's example.py) are moved to the model's GPU for the
call and the result is returned on the caller's device.
"""
import ctypes
import os

import numpy as np
import torch
from torch import nn

from . import _abi, weights
from .config import DEFAULT_CONFIG, load_config

_ROOT = os.path.abspath(os.path.dirname(__file__))
default_config = DEFAULT_CONFIG
# The reference resolves its default checkpoints next to its module file (bvrnn_codec_model.py:11-15: ./chkpts/...).
# Same file names here, looked up in $BVC_CHKPT_DIR if set, else in ./chkpts next to the drop-in shim
# (bvrnn_codec_model.py at the repository root).  The files themselves are Git-LFS objects of the reference and are
# not shipped: the constructor names the path it looked at when one is missing.
_CHKPT_DIR = os.environ.get("BVC_CHKPT_DIR") or os.path.join(os.path.dirname(_ROOT), "chkpts")
default_chkpt_bvrnn = os.path.join(_CHKPT_DIR, "bvrnn_var_bitrate_step200000")
default_chkpt_vocoder = os.path.join(_CHKPT_DIR, "bigvgan_causal_tiny_ftbvrnn_g_step3500000")

SCALING = 10 ** (-10 / 20)      # bvrnn_codec_model.py:17


def _as_device(device):
    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError("bvcodec needs an AMD GPU (gfx950): torch.cuda.is_available() is False and "
                               "there is no CPU fallback")
        return torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError(f"bvcodec runs on the GPU only (got device '{device}')")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


class _Engine:
    """One device-resident copy of the weights (a bvc_model handle) + its workspace."""

    def __init__(self, conf, tensors, device):
        self.lib = _abi.load()
        self.device = device
        self.conf = conf
        v = conf["vocoder_config"]
        cfg = _abi.BvcConfig()
        cfg.num_mels, cfg.h_dim, cfg.z_dim = conf["num_mels"], conf["h_dim"], conf["z_dim"]
        cfg.var_bit = 1 if conf["var_bit"] else 0
        cfg.n_fft, cfg.hop, cfg.pad_left = conf["winsize"], conf["hopsize"], conf["mel_pad_left"]
        cfg.sample_rate, cfg.fmin, cfg.fmax = conf["fs"], float(conf["fmin"]), float(conf["fmax"])
        cfg.upsample_initial_channel = v["upsample_initial_channel"]
        cfg.n_up = len(v["upsample_rates"])
        for i, (u, k) in enumerate(zip(v["upsample_rates"], v["upsample_kernel_sizes"])):
            cfg.up_rates[i], cfg.up_kernels[i] = u, k
        cfg.n_resk = len(v["resblock_kernel_sizes"])
        for j, (k, ds) in enumerate(zip(v["resblock_kernel_sizes"], v["resblock_dilation_sizes"])):
            cfg.res_kernels[j] = k
            for d in range(3):
                cfg.res_dilations[j][d] = ds[d]
        names = list(tensors)
        arr = (_abi.BvcTensor * len(names))()
        for i, n in enumerate(names):
            t = tensors[n]
            arr[i].name, arr[i].h_data, arr[i].numel = n.encode(), t.data_ptr(), t.numel()
        handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _abi.check(self.lib.bvc_model_create(ctypes.byref(cfg), arr, len(names), ctypes.byref(handle)))
        self.handle = handle
        self._ws = {}            # one workspace per stream: calls on different streams may overlap

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.bvc_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def workspace(self, B, T):
        """Scratch for one call, owned by the CURRENT stream (the C ABI allows one in-flight call per
        (model, workspace); independent batches issued on different streams therefore overlap)."""
        need = self.lib.bvc_workspace_bytes(self.handle, B, T)
        key = torch.cuda.current_stream(self.device).cuda_stream
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            self._ws[key] = None
            ws = self._ws[key] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return ctypes.c_void_p(ws.data_ptr()), ws.numel()

    def stream(self):
        return _abi.current_stream(self.device)

    def set_option(self, name, value):
        _abi.check(self.lib.bvc_model_set_option(self.handle, name.encode(), int(value)))

    def get_option(self, name):
        v = ctypes.c_int32(0)
        _abi.check(self.lib.bvc_model_get_option(self.handle, name.encode(), ctypes.byref(v)))
        return int(v.value)

    def check_status(self):
        """Synchronises the device and raises if a persistent recurrence kernel of this model ever gave up waiting
        (its results were then invalid); see bvc_model_status in include/bvcodec.h.  (Every compute call also looks at
        the status word, without synchronising, and raises on the first call after a time-out.)"""
        code = ctypes.c_uint32(0)
        with torch.cuda.device(self.device):
            _abi.check(self.lib.bvc_model_status(self.handle, ctypes.byref(code)))

    def poll_status(self):
        """The same check without synchronising (bvc_model_poll_status): for right after a blocking copy of an output."""
        _abi.check(self.lib.bvc_model_poll_status(self.handle, None))

    def deliver(self, out, out_dev):
        """An output tensor on the caller's device.  A copy to the CPU blocks until the call has finished: the recurrence's status word is
        then final for THIS call, and a time-out is raised here rather than by the next call."""
        res = out.to(out_dev)
        if torch.device(out_dev).type == "cpu":
            self.poll_status()
        return res

    def num_frames(self, L):
        return int(self.lib.bvc_num_frames(self.handle, L))

    def vocoder_length(self, T):
        return int(self.lib.bvc_vocoder_length(self.handle, T))


def _trimmed(total, length):
    """Samples that ``x[:, :, :length]`` keeps of ``total`` (models.py:238): Python slice semantics, so 0 keeps nothing and a
    negative length cuts from the end."""
    return len(range(int(total))[:int(length)])


def _prep(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class _OnDevice(nn.Module):
    """Shared plumbing: lazily creates the engine on the module's device."""

    def __init__(self, conf, tensors):
        super().__init__()
        self.conf = conf
        self._tensors = tensors
        self._engines = {}
        self._device = None

    def _apply(self, fn, *a, **k):              # follow .to('cuda:N') / .cuda(); there is no CPU residence
        # probe where a tensor that lives where this module lives would end up: a dtype-only conversion (.float(), .half(),
        # .to(torch.float32)) leaves the device alone and must not forget the residence
        here = torch.empty(0, device=self._device) if self._device is not None else torch.empty(0)
        probe = fn(here)
        if probe.device.type == "cuda":
            self._device = _as_device(probe.device)
        elif probe.device.type == "cpu" and here.device.type != "cpu":    # .cpu() / .to('cpu'): the next call follows its input again
            self._device = None
        return super()._apply(fn, *a, **k)

    _SCHEDULES = {"persistent": 0, "layers": 1, "auto": 2}

    def set_recurrence(self, schedule):
        """'auto' (default): every frame of a call in one persistent kernel launch while calls come one at a time, one launch
        per layer (hipGraph replay) while calls issued on several streams overlap - the library watches whether the previous
        call, issued on another stream, is still running when a new one starts;
        'persistent' / 'layers': that schedule for every call.  Applies to the engines created so far and to later ones.
        Not while a call is in flight."""
        if schedule not in self._SCHEDULES:
            raise ValueError("schedule must be 'auto', 'persistent' or 'layers'")
        self._recurrence = schedule
        for eng in list(self._engines.values()):
            eng.set_option("recurrence", self._SCHEDULES[schedule])

    def check_status(self):
        """Device-synchronising health check of the engines this module has created (bvc_model_status)."""
        for eng in list(self._engines.values()):
            eng.check_status()

    def engine(self, like=None):
        dev = self._device
        if dev is None and like is not None and like.device.type == "cuda":
            dev = like.device
        dev = _as_device(dev)
        if dev not in self._engines:
            self._engines[dev] = _Engine(self.conf, self._tensors, dev)
            if getattr(self, "_recurrence", None) is not None:
                self._engines[dev].set_option("recurrence", self._SCHEDULES[self._recurrence])
        return self._engines[dev]


class BVRNN(_OnDevice):
    """``BVRNN.encode`` / ``BVRNN.decode`` (bvrnn.py:163-229) on the HIP recurrent kernels."""

    def __init__(self, conf, tensors, shared=None):
        super().__init__(conf, tensors)
        self.h_dim, self.z_dim, self.x_dim = conf["h_dim"], conf["z_dim"], conf["num_mels"]
        self.varBit = bool(conf["var_bit"])
        if shared is not None:
            self._engines = shared

    @torch.no_grad()
    def forward(self, y, p_use_gen, greedy, varBitrate, *, r=None, noise=None, return_all=False):
        """``BVRNN.forward(y, p_use_gen, greedy, varBitrate)`` (bvrnn.py:86-160), forward values only (this build
        has no autograd): stochastic Bernoulli sampler ``round(u - 0.5 + enc_t)``, prior net, KL term and the
        teacher-forcing mix.  Returns (all_dec_mean (B,T,x_dim), kld scalar) like the reference.

        Randomness: the reference draws, per frame, ``torch.rand([])`` (compared with p_use_gen) and - unless
        greedy - ``torch.rand_like(enc_t)``.  By default the same numbers are drawn here, in the same order, from
        torch's global CPU generator, so ``torch.manual_seed(s)`` followed by this call reproduces the reference
        run on CPU after the same seed.  ``r`` (T,) / ``noise`` (B,T,z_dim) override them.
        ``return_all=True`` adds a dict with z (forward value of the sample), prob (enc_t), prior, kld_frames."""
        eng = self.engine(y)
        out_dev = y.device
        y = _prep(y, eng.device)
        B, T, _ = y.shape
        if r is None or (noise is None and not greedy):
            rs, ns = [], []
            for _ in range(T):                                    # the reference's draw order (bvrnn.py:111,126)
                rs.append(torch.rand([]))
                if not greedy:
                    ns.append(torch.rand(B, self.z_dim))
            r = torch.stack(rs) if r is None else r
            if not greedy and noise is None:
                noise = torch.stack(ns, 1)
        use_gen = (torch.as_tensor(r).detach().cpu().float() < p_use_gen).to(torch.uint8).contiguous()
        assert use_gen.numel() == T
        bits = _prep(varBitrate, eng.device) if (varBitrate is not None and self.varBit) else None
        un = None if greedy else _prep(noise, eng.device)
        dec = torch.empty(B, T, self.x_dim, device=eng.device)
        kld = torch.empty(T, device=eng.device)
        extra = {k: torch.empty(B, T, self.z_dim, device=eng.device) for k in ("z", "prob", "prior")} if return_all else {}
        ws, nws = eng.workspace(B, T)
        with torch.cuda.device(eng.device):
            _abi.check(eng.lib.bvc_bvrnn_forward(
                eng.handle, _abi.ptr(y), _abi.ptr(bits), ctypes.c_void_p(use_gen.data_ptr()), int(p_use_gen < 1),
                int(p_use_gen > 0), _abi.ptr(un), B, T, _abi.ptr(dec), _abi.ptr(kld), _abi.ptr(extra.get("z")),
                _abi.ptr(extra.get("prob")), _abi.ptr(extra.get("prior")), ws, nws, eng.stream()))
        out = (dec.to(out_dev), torch.mean(kld).to(out_dev))     # torch.mean(torch.stack(kld_loss)), bvrnn.py:160
        if return_all:
            extra["kld_frames"] = kld
            return out + ({k: v.to(out_dev) for k, v in extra.items()},)
        return out

    @torch.no_grad()
    def encode(self, y, varBitrate, h, return_prob=False):
        """y (B,T,x_dim), varBitrate (B,T) bits/frame, h (1,B,h_dim) -> (codes (B,T,z), all_h (B,T,h))."""
        eng = self.engine(y)
        out_dev = y.device
        y = _prep(y, eng.device)
        B, T, _ = y.shape
        bits = _prep(varBitrate, eng.device) if varBitrate is not None else None
        h0 = _prep(h.reshape(B, self.h_dim), eng.device)
        codes = torch.empty(B, T, self.z_dim, device=eng.device)
        all_h = torch.empty(B, T, self.h_dim, device=eng.device)
        prob = torch.empty(B, T, self.z_dim, device=eng.device) if return_prob else None
        ws, nws = eng.workspace(B, T)
        with torch.cuda.device(eng.device):
            _abi.check(eng.lib.bvc_bvrnn_encode(eng.handle, _abi.ptr(y), _abi.ptr(bits), _abi.ptr(h0), B, T,
                                                _abi.ptr(codes), _abi.ptr(all_h), None, _abi.ptr(prob), ws, nws,
                                                eng.stream()))
        if return_prob:
            return codes.to(out_dev), all_h.to(out_dev), eng.deliver(prob, out_dev)
        return codes.to(out_dev), eng.deliver(all_h, out_dev)

    @torch.no_grad()
    def encode_stateful(self, y, varBitrate, h):
        """Like encode() but returns the state AFTER the last frame instead of the per-frame states:
        (codes (B,T,z), h_next (1,B,h_dim)).  This is what chunked / streaming encoding needs (the
        reference's encode() only exposes the state before each frame, bvrnn.py:205,209)."""
        eng = self.engine(y)
        out_dev = y.device
        y = _prep(y, eng.device)
        B, T, _ = y.shape
        bits = _prep(varBitrate, eng.device) if varBitrate is not None else None
        h0 = _prep(h.reshape(B, self.h_dim), eng.device)
        codes = torch.empty(B, T, self.z_dim, device=eng.device)
        hT = torch.empty(B, self.h_dim, device=eng.device)
        ws, nws = eng.workspace(B, T)
        with torch.cuda.device(eng.device):
            _abi.check(eng.lib.bvc_bvrnn_encode(eng.handle, _abi.ptr(y), _abi.ptr(bits), _abi.ptr(h0), B, T,
                                                _abi.ptr(codes), None, _abi.ptr(hT), None, ws, nws, eng.stream()))
        return codes.to(out_dev), eng.deliver(hT.unsqueeze(0), out_dev)

    @torch.no_grad()
    def decode(self, z, h):
        """z (B,T,z_dim), h (1,B,h_dim) -> (mel (B,T,x_dim), h (1,B,h_dim))."""
        eng = self.engine(z)
        out_dev = z.device
        z = _prep(z, eng.device)
        B, T, _ = z.shape
        h0 = _prep(h.reshape(B, self.h_dim), eng.device)
        mel = torch.empty(B, T, self.x_dim, device=eng.device)
        hT = torch.empty(B, self.h_dim, device=eng.device)
        ws, nws = eng.workspace(B, T)
        with torch.cuda.device(eng.device):
            _abi.check(eng.lib.bvc_bvrnn_decode(eng.handle, _abi.ptr(z), _abi.ptr(h0), B, T, _abi.ptr(mel),
                                                _abi.ptr(hT), ws, nws, eng.stream()))
        return mel.to(out_dev), eng.deliver(hT.unsqueeze(0), out_dev)


class BigVGAN(_OnDevice):
    """``BigVGAN.forward(x, length)`` (models.py:207-238): x (B, num_mels, T) -> (B, 1, n)."""

    def __init__(self, conf, tensors, shared=None):
        super().__init__(conf, tensors)
        if shared is not None:
            self._engines = shared

    @torch.no_grad()
    def forward(self, x, length, _scale_div=1.0, _time_major=False):
        eng = self.engine(x)
        out_dev = x.device
        mel = _prep(x if _time_major else x.permute(0, 2, 1), eng.device)        # kernels are time-major
        B, T, _ = mel.shape
        n = _trimmed(eng.vocoder_length(T), length)
        wav = torch.empty(B, n, device=eng.device)
        if n:
            ws, nws = eng.workspace(B, T)
            with torch.cuda.device(eng.device):
                _abi.check(eng.lib.bvc_bigvgan(eng.handle, _abi.ptr(mel), B, T, n, float(_scale_div),
                                               _abi.ptr(wav), ws, nws, eng.stream()))
        return wav.unsqueeze(1).to(out_dev)


class BVRNNCodecModel(_OnDevice):
    def __init__(self, config_path=default_config, bvrnn_chkpt_path=default_chkpt_bvrnn,
                 vocoder_chkpt_path=default_chkpt_vocoder):
        """Same three arguments as the reference constructor (bvrnn_codec_model.py:20-42): the TOML file that describes
        front-end, coder and vocoder, and the two ``torch.save`` dicts holding the coder's ``'vrnn'`` and the vocoder's
        ``'generator'`` state dict.  Both checkpoints are loaded strictly; nothing touches the GPU until the first call."""
        for what, path in (("BVRNN", bvrnn_chkpt_path), ("vocoder", vocoder_chkpt_path)):
            if not os.path.exists(path):
                raise FileNotFoundError(f"{what} checkpoint not found at '{path}'.  Pass the path explicitly, or place the "
                                        f"reference's chkpts/ files (Git-LFS objects, not shipped) in '{_CHKPT_DIR}' "
                                        "(override with BVC_CHKPT_DIR)")
        conf = load_config(config_path)
        vrnn_sd = weights.load_checkpoint(bvrnn_chkpt_path, "vrnn")
        gen_sd = weights.load_checkpoint(vocoder_chkpt_path, "generator")
        super().__init__(conf, weights.host_tensors(conf, vrnn_sd, gen_sd))
        _abi.load()                                   # fail at construction if the HIP library is absent
        self.bvrnn = BVRNN(conf, self._tensors, shared=self._engines)
        self.vocoder = BigVGAN(conf, self._tensors, shared=self._engines)

    def bits_per_frame(self, bitrate):
        return float(np.round(bitrate * self.conf['hopsize'] / self.conf['fs']))   # bvrnn_codec_model.py:58

    @torch.no_grad()
    def mel_spectrogram(self, x, scale=SCALING):
        """log-mel of x*scale as the facade computes it (bvrnn_codec_model.py:49-56): (B,L) -> (B,T,80)."""
        eng = self.engine(x)
        out_dev = x.device
        x = _prep(x, eng.device)
        if x.dim() != 2:
            raise RuntimeError("expected a (batch, length) waveform")
        B, L = x.shape
        T = eng.num_frames(L)
        if T <= 0:
            raise RuntimeError(f"Argument #4: Padding size should be less than the corresponding input dimension "
                               f"(length {L} is too short for the reflect padding of the STFT front-end)")
        mel = torch.empty(B, T, self.conf["num_mels"], device=eng.device)
        with torch.cuda.device(eng.device):
            _abi.check(eng.lib.bvc_stft_logmel(eng.handle, _abi.ptr(x), B, L, float(scale), _abi.ptr(mel),
                                               eng.stream()))
        return mel.to(out_dev)

    @torch.no_grad()
    def encode(self, x, bitrate):
        """Waveforms ``x`` (batch, samples), expected in [-1, 1], to codes (batch, samples // hop, z_dim) with values in
        {0, 1} and 0.5 at masked positions.  ``bitrate`` [bit/s] selects round(bitrate * hop / fs) leading bits per frame
        (saturating at z_dim; ignored by fixed-rate configs) - bvrnn_codec_model.py:44-62."""
        eng = self.engine(x)
        out_dev = x.device
        x = _prep(x, eng.device)
        if x.dim() != 2:
            raise RuntimeError("expected a (batch, length) waveform")
        B, L = x.shape
        T = eng.num_frames(L)
        if T <= 0:
            raise RuntimeError(f"Argument #4: Padding size should be less than the corresponding input dimension "
                               f"(length {L} is too short for the reflect padding of the STFT front-end)")
        codes = torch.empty(B, T, self.conf["z_dim"], device=eng.device)
        ws, nws = eng.workspace(B, T)
        with torch.cuda.device(eng.device):
            _abi.check(eng.lib.bvc_encode(eng.handle, _abi.ptr(x), B, L, float(SCALING), self.bits_per_frame(bitrate),
                                          _abi.ptr(codes), ws, nws, eng.stream()))
        return eng.deliver(codes, out_dev)

    @torch.no_grad()
    def decode(self, codes, length):
        """Codes (batch, frames, z_dim) back to waveforms (batch, min(length, 256 * frames + 294)): coder decode from a
        zero state, vocoder, output gain undone - bvrnn_codec_model.py:64-71."""
        eng = self.engine(codes)
        out_dev = codes.device
        codes = _prep(codes, eng.device)
        B, T, Z = codes.shape
        if Z != self.conf["z_dim"]:
            raise RuntimeError(f"codes must have {self.conf['z_dim']} values per frame, got {Z}")
        n = _trimmed(eng.vocoder_length(T), length)       # `[:, :, :length]`, models.py:238
        wav = torch.empty(B, n, device=eng.device)
        if n:                                             # (length 0 keeps nothing: an empty (B, 0) tensor like the reference's)
            ws, nws = eng.workspace(B, T)
            with torch.cuda.device(eng.device):
                _abi.check(eng.lib.bvc_decode(eng.handle, _abi.ptr(codes), B, T, n, float(SCALING),
                                              _abi.ptr(wav), ws, nws, eng.stream()))
        return eng.deliver(wav, out_dev)

    def forward(self, x, bitrate):
        """decode(encode(x, bitrate)) trimmed to the input length (bvrnn_codec_model.py:73-76)."""
        length = x.shape[1]
        codes = self.encode(x, bitrate)
        return self.decode(codes, length)

    @torch.no_grad()
    def forward_fused(self, x, bitrate, return_codes=False):
        """``forward(x, bitrate)`` with ONE recurrence instead of two (``bvc_forward``): the encoder's frame loop already runs the
        decoder on every frame (bvrnn.py:198-204) from the states ``decode`` would visit again, so its outputs go to the vocoder
        directly.  Same codes as ``encode``; the waveform agrees with ``forward`` to rounding (the halves of dec.0 and of the GRU's
        input gates are summed in another order).  ``forward`` itself stays the reference's decode(encode(x))."""
        eng = self.engine(x)
        out_dev = x.device
        x = _prep(x, eng.device)
        if x.dim() != 2:
            raise RuntimeError("expected a (batch, length) waveform")
        B, L = x.shape
        T = eng.num_frames(L)
        if T <= 0:
            raise RuntimeError(f"Argument #4: Padding size should be less than the corresponding input dimension "
                               f"(length {L} is too short for the reflect padding of the STFT front-end)")
        n = _trimmed(eng.vocoder_length(T), L)
        wav = torch.empty(B, n, device=eng.device)
        codes = torch.empty(B, T, self.conf["z_dim"], device=eng.device) if return_codes else None
        ws, nws = eng.workspace(B, T)
        with torch.cuda.device(eng.device):
            _abi.check(eng.lib.bvc_forward(eng.handle, _abi.ptr(x), B, L, float(SCALING), self.bits_per_frame(bitrate), n, float(SCALING),
                                           _abi.ptr(codes), _abi.ptr(wav), ws, nws, eng.stream()))
        if return_codes:
            return codes.to(out_dev), eng.deliver(wav, out_dev)
        return eng.deliver(wav, out_dev)

    # ---- wire format (not in the reference, which has no bit stream: SURVEY.md 8f rank 2)
    def active_bits(self, bitrate):
        z = self.conf["z_dim"]
        return z if not self.conf["var_bit"] else int(min(z, max(0.0, self.bits_per_frame(bitrate))))

    @torch.no_grad()
    def pack(self, codes, bitrate):
        """codes (B,T,z_dim) from encode(x, bitrate) -> uint8 (B,T,ceil(n/8)), n active bits per frame."""
        eng = self.engine(codes)
        out_dev = codes.device
        codes = _prep(codes, eng.device)
        B, T, Z = codes.shape
        n = self.active_bits(bitrate)
        out = torch.empty(B, T, (n + 7) // 8, dtype=torch.uint8, device=eng.device)
        if out.numel():
            with torch.cuda.device(eng.device):
                _abi.check(eng.lib.bvc_pack_codes(_abi.ptr(codes), B, T, Z, n, ctypes.c_void_p(out.data_ptr()),
                                                  eng.stream()))
        return out.to(out_dev)

    @torch.no_grad()
    def unpack(self, packed, bitrate):
        """Inverse of pack(): uint8 (B,T,ceil(n/8)) -> float32 codes (B,T,z_dim) with 0.5 in masked positions."""
        eng = self.engine(packed)
        out_dev = packed.device
        packed = packed.to(eng.device).contiguous()
        B, T, _ = packed.shape
        n = self.active_bits(bitrate)
        Z = self.conf["z_dim"]
        codes = torch.empty(B, T, Z, device=eng.device)
        with torch.cuda.device(eng.device):
            _abi.check(eng.lib.bvc_unpack_codes(ctypes.c_void_p(packed.data_ptr()) if packed.numel() else None, B, T, Z,
                                                n, _abi.ptr(codes), eng.stream()))
        return codes.to(out_dev)
