"""CPU-side checks of the C-ABI library and the host logic (no GPU, no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_symbol_the_header_declares():
    from bvcodec import _abi
    hdr = open(os.path.join(ROOT, "include", "bvcodec.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(bvc_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 18
    lib = _abi.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/bvcodec.h but not exported"
    assert declared == set(_abi.SIGNATURES), declared ^ set(_abi.SIGNATURES)
    assert lib.bvc_abi_version() == 3


def test_header_is_plain_c_and_links_from_a_c_program(tmp_path):
    """include/bvcodec.h is the drop-in boundary: it must compile as C99 and as C++ with nothing but the standard headers
    (no torch / HIP types in the signatures), and a C program must link against the shared library and get an answer
    from an entry point that needs no device."""
    import shutil, subprocess
    from bvcodec import _abi
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "client.c"
    src.write_text("""#include <stdio.h>
#include "bvcodec.h"
int main(void) {
    bvc_config cfg;
    (void)cfg;
    printf("%d %lld %s\\n", (int)bvc_abi_version(), (long long)bvc_workspace_bytes(0, 4, 10), bvc_last_error() ? "err" : "null");
    return 0;
}
""")
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", inc, str(src)])
    subprocess.check_call(["g++", "-std=c++11", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", "-I", inc, str(src)])
    lib = _abi.LIB_PATH
    exe = tmp_path / "client"
    subprocess.check_call(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), lib, f"-Wl,-rpath,{os.path.dirname(lib)}",
                           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-lstdc++"])
    out = subprocess.check_output([str(exe)], text=True).split()
    assert out[0] == "3" and out[1] == "0"


def test_config_struct_layout_matches_header():
    from bvcodec import _abi
    # 8 int32 + 2 float + 2 int32 + 8 + 8 int32 + 1 + 4 + 12 int32
    assert ctypes.sizeof(_abi.BvcConfig) == 4 * (8 + 2 + 2 + 8 + 8 + 1 + 4 + 12)
    assert ctypes.sizeof(_abi.BvcTensor) == 24


def test_model_create_fails_loudly_without_gpu(conf_var):
    """No CPU fallback: without a device the library refuses to build a model."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bvcodec import _abi
    lib = _abi.load()
    cfg = _abi.BvcConfig()
    cfg.num_mels, cfg.h_dim, cfg.z_dim, cfg.var_bit = 80, 1024, 64, 1
    cfg.n_fft, cfg.hop, cfg.pad_left, cfg.sample_rate = 1024, 256, 256, 22050
    cfg.upsample_initial_channel, cfg.n_up, cfg.n_resk = 128, 4, 3
    for i, (u, k) in enumerate(((8, 16), (8, 16), (2, 4), (2, 4))):
        cfg.up_rates[i], cfg.up_kernels[i] = u, k
    t = (_abi.BvcTensor * 1)()
    dummy = np.zeros(4, dtype=np.float32)
    t[0].name, t[0].h_data, t[0].numel = b"mean_mel", dummy.ctypes.data, 4
    h = ctypes.c_void_p()
    rc = lib.bvc_model_create(ctypes.byref(cfg), t, 1, ctypes.byref(h))
    assert rc == -5 and b"no HIP device" in lib.bvc_last_error()        # BVC_ENODEVICE
    cfg.hop = 128
    assert lib.bvc_model_create(ctypes.byref(cfg), t, 1, ctypes.byref(h)) == -1   # BVC_EINVAL before anything else
    assert lib.bvc_workspace_bytes(None, 4, 10) == 0


def test_facade_constructs_on_cpu_and_refuses_to_compute(tmp_path, conf_var):
    from bvcodec import BVRNNCodecModel, config, synth
    p1, p2 = synth.write_checkpoints(conf_var, str(tmp_path), seed=7)
    m = BVRNNCodecModel(config.DEFAULT_CONFIG, p1, p2)
    m.eval()
    assert m.bits_per_frame(3000) == 35.0 and m.bits_per_frame(1500) == 17.0 and m.bits_per_frame(6000) == 70.0
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m.encode(torch.zeros(1, 4096), 3000)


def test_checkpoint_loading_is_strict(tmp_path, conf_var):
    from bvcodec import BVRNNCodecModel, config, synth
    p1, p2 = synth.write_checkpoints(conf_var, str(tmp_path), seed=7)
    sd = torch.load(p1, weights_only=True)
    del sd["vrnn"]["enc.2.bias"]
    bad = str(tmp_path / "bad")
    torch.save(sd, bad)
    with pytest.raises(RuntimeError, match="Missing key"):
        BVRNNCodecModel(config.DEFAULT_CONFIG, bad, p2)
    with pytest.raises(KeyError):
        BVRNNCodecModel(config.DEFAULT_CONFIG, p2, p2)            # wrong top-level key
    with pytest.raises(FileNotFoundError):
        BVRNNCodecModel(config.DEFAULT_CONFIG, str(tmp_path / "nope"), p2)
    with pytest.raises(FileNotFoundError):
        BVRNNCodecModel()                                         # the shipped checkpoints are LFS pointers


def test_weight_norm_fold_matches_oracle(conf_var):
    from bvcodec import synth, weights
    from oracle import bigvgan as obig
    g = synth.generator_state_dict(conf_var, 3)
    v = synth.bvrnn_state_dict(conf_var, 3)
    ht = weights.host_tensors(conf_var, v, g)
    for name in ("conv_pre", "ups.1.1", "resblocks.4.convs1.2", "conv_post"):
        ref = obig.fold_weight_norm(g[name + ".weight_g"], g[name + ".weight_v"])
        assert torch.equal(ht[name + ".weight"], ref) or (ht[name + ".weight"] - ref).abs().max() < 1e-7
    assert "prior.0.weight" in ht and "log_sigma" not in ht       # prior net: bvc_bvrnn_forward; log_sigma: loss only
    assert ht["mel_basis"].shape == (80, 513) and ht["hann_window"].shape == (1024,)
    assert all(t.dtype == torch.float32 and t.is_contiguous() for t in ht.values())


def test_product_melbank_equals_oracle_melbank():
    from bvcodec import melbank
    from oracle import melbank as omel
    assert np.array_equal(melbank.slaney_mel_basis(22050, 1024, 80, 0, 8000), omel.mel_filterbank(22050, 1024, 80, 0, 8000))


def test_config_validation(tmp_path):
    from bvcodec import config
    ref_like = open(config.DEFAULT_CONFIG).read()
    p = tmp_path / "c.toml"
    p.write_text(ref_like.replace('activation = "snakebeta"', 'activation = "lrelu"'))
    with pytest.raises(ValueError, match="snakebeta"):
        config.load_config(str(p))
    p.write_text(ref_like.replace("hopsize = 256", "hopsize = 128"))
    with pytest.raises(ValueError):
        config.load_config(str(p))
    p.write_text(ref_like.replace("h_dim = 1024\n", ""))
    with pytest.raises(KeyError):
        config.load_config(str(p))
    c64 = config.load_config(config.DEFAULT_CONFIG_64BIT)
    assert c64["var_bit"] is False


def test_resampler_filter_design_matches_scipy():
    """The restated Kaiser low-pass / padding of scipy.signal.resample_poly (example.py:15)."""
    import scipy.signal as ss
    from bvcodec import preprocess
    h = preprocess.kaiser_lowpass(3201, 1.0 / 160)
    assert np.abs(h - ss.firwin(3201, 1.0 / 160, window=("kaiser", 5.0))).max() < 1e-15
    up, down, hp, npre = preprocess.design(22050, 24000)
    assert (up, down) == (147, 160) and npre == 11 and len(hp) == 3201 + 160
    # numpy emulation of the GPU kernel's index walk == scipy on a small signal
    rng = np.random.default_rng(0)
    x = rng.standard_normal(700)
    ref = ss.resample_poly(x, 22050, 24000)
    n_out = len(ref)
    y = np.zeros(n_out)
    for m in range(n_out):
        pos = (m + npre) * down
        i, k = pos // up, pos % up
        if i >= len(x):
            k += (i - (len(x) - 1)) * up
            i = len(x) - 1
        acc = 0.0
        while k < len(hp) and i >= 0:
            acc += hp[k] * x[i]
            k += up
            i -= 1
        y[m] = acc
    assert np.abs(y - ref).max() < 1e-12
