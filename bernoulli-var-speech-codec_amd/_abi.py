"""ctypes binding of csrc/libbvcodec_hip.so (C ABI: include/bvcodec.h).

Thin by design: tensors cross the boundary as raw device pointers (``tensor.data_ptr()``), the
stream as ``torch.cuda.current_stream().cuda_stream``.  There is NO fallback: if the library is not
built, or no GPU is visible when a model is created, the error is raised to the caller.
"""
import ctypes
import os

import torch  # noqa: F401  (loads torch's libamdhip64 first so the library binds to the same HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BVC_LIB") or os.path.join(_HERE, "csrc", "libbvcodec_hip.so")   # BVC_LIB: experiments only

c_float_p = ctypes.POINTER(ctypes.c_float)


class BvcConfig(ctypes.Structure):
    _fields_ = [
        ("num_mels", ctypes.c_int32), ("h_dim", ctypes.c_int32), ("z_dim", ctypes.c_int32),
        ("var_bit", ctypes.c_int32), ("n_fft", ctypes.c_int32), ("hop", ctypes.c_int32),
        ("pad_left", ctypes.c_int32), ("sample_rate", ctypes.c_int32),
        ("fmin", ctypes.c_float), ("fmax", ctypes.c_float),
        ("upsample_initial_channel", ctypes.c_int32), ("n_up", ctypes.c_int32),
        ("up_rates", ctypes.c_int32 * 8), ("up_kernels", ctypes.c_int32 * 8),
        ("n_resk", ctypes.c_int32), ("res_kernels", ctypes.c_int32 * 4),
        ("res_dilations", (ctypes.c_int32 * 3) * 4),
    ]


class BvcTensor(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("h_data", ctypes.c_void_p), ("numel", ctypes.c_int64)]


# name -> (restype, argtypes); every symbol include/bvcodec.h declares
_vp, _i32, _i64, _f, _sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t
SIGNATURES = {
    "bvc_abi_version": (ctypes.c_int, []),
    "bvc_last_error": (ctypes.c_char_p, []),
    "bvc_model_create": (ctypes.c_int, [ctypes.POINTER(BvcConfig), ctypes.POINTER(BvcTensor), _i32,
                                        ctypes.POINTER(_vp)]),
    "bvc_model_destroy": (None, [_vp]),
    "bvc_model_set_option": (ctypes.c_int, [_vp, ctypes.c_char_p, _i32]),
    "bvc_model_get_option": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.POINTER(_i32)]),
    "bvc_flow_fence": (ctypes.c_int, [_vp]),
    "bvc_model_status": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint32)]),
    "bvc_model_poll_status": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint32)]),
    "bvc_num_frames": (_i64, [_vp, _i64]),
    "bvc_vocoder_length": (_i64, [_vp, _i64]),
    "bvc_workspace_bytes": (_sz, [_vp, _i32, _i64]),
    "bvc_stft_logmel": (ctypes.c_int, [_vp, _vp, _i32, _i64, _f, _vp, _vp]),
    "bvc_bvrnn_encode": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "bvc_bvrnn_decode": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _sz, _vp]),
    "bvc_bvrnn_forward": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp,
                                         _vp, _sz, _vp]),
    "bvc_bigvgan": (ctypes.c_int, [_vp, _vp, _i32, _i64, _i64, _f, _vp, _vp, _sz, _vp]),
    "bvc_vocoder_stream_create": (ctypes.c_int, [_vp, _i32, _i32, ctypes.POINTER(_vp)]),
    "bvc_vocoder_stream_destroy": (None, [_vp]),
    "bvc_vocoder_stream_reset": (ctypes.c_int, [_vp, _vp]),
    "bvc_vocoder_stream_push": (ctypes.c_int, [_vp, _vp, _i32, _f, _vp, _vp]),
    "bvc_stream_codec_create": (ctypes.c_int, [_vp, _i32, _i32, _f, _f, _f, ctypes.POINTER(_vp)]),
    "bvc_stream_codec_destroy": (None, [_vp]),
    "bvc_stream_codec_buffers": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_i32)]),
    "bvc_stream_codec_tick": (ctypes.c_int, [_vp, ctypes.POINTER(_i32), _vp]),
    "bvc_encode": (ctypes.c_int, [_vp, _vp, _i32, _i64, _f, _f, _vp, _vp, _sz, _vp]),
    "bvc_decode": (ctypes.c_int, [_vp, _vp, _i32, _i64, _i64, _f, _vp, _vp, _sz, _vp]),
    "bvc_forward": (ctypes.c_int, [_vp, _vp, _i32, _i64, _f, _f, _i64, _f, _vp, _vp, _vp, _sz, _vp]),
    "bvc_resample_poly": (ctypes.c_int, [_vp, _i32, _i64, _vp, _i32, _i32, _i32, _i64, _vp, _i64, _vp]),
    "bvc_peak_normalize": (ctypes.c_int, [_vp, _i32, _i64, _vp]),
    "bvc_pack_codes": (ctypes.c_int, [_vp, _i32, _i64, _i32, _i32, _vp, _vp]),
    "bvc_unpack_codes": (ctypes.c_int, [_vp, _i32, _i64, _i32, _i32, _vp, _vp]),
    "bvc_probe_begin": (ctypes.c_int, [_i32, _i32, _i32]),
    "bvc_probe_end": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                     ctypes.POINTER(_i32)]),
    "bvc_kprobe_enable": (ctypes.c_int, [_i32]),
    "bvc_kprobe_read": (ctypes.c_int, [_i32, _i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                       ctypes.POINTER(_i32)]),
    "bvc_kprobe_read_span": (ctypes.c_int, [_i32, _i32, _i32, _i32, ctypes.POINTER(ctypes.c_double),
                                            ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i32)]),
    "bvc_test_linear": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "bvc_test_linear_batched": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "bvc_test_snakebeta": (ctypes.c_int, [_vp, _i64, _f, _f, _vp, _vp]),
    "bvc_test_vocoder_tap": (ctypes.c_int, [_vp, _vp, _i32, _i64, _i32, _vp, ctypes.POINTER(_i64), _vp, _sz, _vp]),
}

_lib = None


class BvcError(RuntimeError):
    pass


def load():
    """dlopen the HIP library (once).  Raises if it has not been built - no CPU fallback exists."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BvcError(f"{LIB_PATH} is missing: build it with `python -m bvcodec.build` "
                           "(or __graft_entry__.build()); this package has no CPU fallback")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.bvc_abi_version() != 3:
            raise BvcError("libbvcodec_hip.so ABI version mismatch")
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        msg = load().bvc_last_error()
        raise BvcError(f"bvcodec error {rc}: {msg.decode() if msg else ''}")


def ptr(t):
    """Device (or host) pointer of a contiguous float32 tensor, or None."""
    if t is None:
        return None
    assert t.dtype == torch.float32 and t.is_contiguous(), "float32 contiguous tensors only"
    return ctypes.c_void_p(t.data_ptr())


def current_stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
