#!/usr/bin/env python3
"""Non-asserting diagnostic sweep of every stage of the HIP path against goldens / oracle.
Prints one line per check; used while bringing kernels up (python tests/gpu_diag.py)."""
import ctypes
import os
import sys
import time
import traceback

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from conftest import load_golden          # noqa: E402
from gpu_common import make_model, report  # noqa: E402
from bvcodec import _abi, synth           # noqa: E402

DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.asarray(a))


def section(name, fn):
    print(f"==== {name}", flush=True)
    try:
        fn()
    except Exception:
        traceback.print_exc()
        sys.stdout.flush()


def gemm():
    lib = _abi.load()
    for (M, N, K) in [(64, 1024, 1024), (7, 64, 1024), (100, 1024, 80), (64, 1024, 2048)]:
        g = torch.Generator().manual_seed(1)
        x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / np.sqrt(K); b = torch.randn(N, generator=g)
        ref = torch.nn.functional.elu(torch.nn.functional.linear(x.double(), w.double(), b.double()))
        xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
        for name, fn in (("skinny", lib.bvc_test_linear), ("batched", lib.bvc_test_linear_batched)):
            y = torch.full((M, N), float("nan"), device=DEV)
            _abi.check(fn(_abi.ptr(xd), _abi.ptr(wd), _abi.ptr(bd), M, N, K, 1, _abi.ptr(y),
                          _abi.current_stream(torch.device(DEV))))
            torch.cuda.synchronize()
            report(f"gemm {name} M{M} N{N} K{K}", y.cpu().numpy(), ref.numpy())


def frontend():
    model = make_model()[0]
    for name in ("g1_mel", "g1_mel_short"):
        g = load_golden(name)
        mel = model.mel_spectrogram(t(g["x"]).to(DEV)).cpu().numpy()
        ref = np.transpose(g["mel"], (0, 2, 1))
        for b in range(mel.shape[0]):
            report(f"mel {name}[{b}] log", mel[b], ref[b])
            report(f"mel {name}[{b}] lin", np.exp(mel[b]), np.exp(ref[b]))


def bvrnn():
    for tag, h, vb in (("h64_var", 64, True), ("h1024_var", 1024, True), ("h1024_fix", 1024, False)):
        model = make_model(vb, h)[0]
        g = load_golden(f"g3_bvrnn_{tag}")
        B = g["y"].shape[0]
        h0 = torch.zeros(1, B, h, device=DEV)
        codes, all_h, prob = model.bvrnn.encode(t(g["y"]).to(DEV), t(g["bits"]).to(DEV), h0, return_prob=True)
        report(f"bvrnn {tag} prob", prob.cpu().numpy(), g["prob"])
        report(f"bvrnn {tag} all_h", all_h.cpu().numpy(), g["all_h"])
        d = codes.cpu().numpy() != g["codes"]
        print(f"bvrnn {tag} code mismatches {int(d.sum())} of {d.size}; first frame with mismatch "
              f"{int(np.argmax(d.any(axis=(0, 2)))) if d.any() else -1}")
        mel, hT = model.bvrnn.decode(t(g["codes"]).to(DEV), h0)
        report(f"bvrnn {tag} mel_hat", mel.cpu().numpy(), g["mel_hat"])
        report(f"bvrnn {tag} h_T", hT[0].cpu().numpy(), g["h_T"])


def vocoder():
    lib = _abi.load()
    model = make_model()[0]
    g = load_golden("g5_bigvgan_taps")
    mel = t(g["mel"]).permute(0, 2, 1).contiguous().to(DEV)
    B, T = mel.shape[0], mel.shape[1]
    eng = model.engine(mel)
    ws, nws = eng.workspace(B, T)
    names = ["conv_pre"]
    for i in range(4):
        names += [f"up{i}", f"stage{i}"]
    for which, nm in enumerate(names):
        n = ctypes.c_int64()
        _abi.check(lib.bvc_test_vocoder_tap(eng.handle, _abi.ptr(mel), B, T, which, None, ctypes.byref(n), ws, nws, eng.stream()))
        out = torch.full((B, n.value), float("nan"), device=DEV)
        _abi.check(lib.bvc_test_vocoder_tap(eng.handle, _abi.ptr(mel), B, T, which, _abi.ptr(out), ctypes.byref(n), ws, nws, eng.stream()))
        torch.cuda.synchronize()
        if nm.startswith("stage"):
            i = int(nm[5:]); ref = (g[f"res{i}_0"] + g[f"res{i}_1"] + g[f"res{i}_2"]) / 3
        else:
            ref = g[nm]
        got = out.cpu().numpy().reshape(B, -1, ref.shape[1]).transpose(0, 2, 1)
        if got.shape != ref.shape:
            print(f"tap {nm}: SHAPE {got.shape} vs {ref.shape}")
            continue
        report(f"tap {nm}", got, ref)
    w = model.vocoder(t(g["mel"]).to(DEV), 10 ** 9).cpu().numpy()
    report("vocoder wav (taps case)", w, g["wav"])
    g2 = load_golden("g5_bigvgan")
    for length in (8192, 8392, 10 ** 9):
        w = model.vocoder(t(g2["mel"]).to(DEV), length).cpu().numpy()
        if w.shape != g2[f"wav_{length}"].shape:
            print("vocoder SHAPE", w.shape, g2[f"wav_{length}"].shape)
        else:
            report(f"vocoder wav len={length}", w, g2[f"wav_{length}"])


def facade():
    for tag in ("var", "fix"):
        model = make_model(tag == "var", 1024)[0]
        g = load_golden(f"g6_e2e_{tag}")
        x = t(g["x"]).to(DEV)
        for br in ((3000, 1500, 6000) if tag == "var" else (3000,)):
            codes = model.encode(x, br).cpu().numpy()
            d = codes != g[f"codes_{br}"]
            print(f"facade {tag} {br}: code mismatches {int(d.sum())} of {d.size}")
            if d.any():
                idx = np.argwhere(d)[:5]
                for (b, tt, n) in idx:
                    print(f"    b{b} t{tt} bit{n}: ref prob {g[f'prob_{br}'][b, tt, n]:.9f}")
            wav = model.decode(t(g[f"codes_{br}"]).to(DEV), x.shape[1]).cpu().numpy()
            report(f"facade {tag} {br} wav", wav, g[f"wav_{br}"])


def timing():
    model = make_model()[0]
    x = synth.synthetic_speech(64, 110250, seed=0, kind="noise").to(DEV)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        codes = model.encode(x, 3000)
        torch.cuda.synchronize(); t1 = time.time()
        wav = model.decode(codes, x.shape[1])
        torch.cuda.synchronize(); t2 = time.time()
        print(f"C2 64x5s: encode {1e3 * (t1 - t0):.1f} ms, decode {1e3 * (t2 - t1):.1f} ms -> "
              f"{64 * 5 / (t2 - t0):.0f} x real-time", flush=True)
    print("finite", bool(torch.isfinite(wav).all()), "wav rms", float(wav.pow(2).mean().sqrt()))



def kprobe():
    """Per-layer in-kernel durations of the graph-replayed recurrent kernels at the C2 shape."""
    lib = _abi.load()
    model = make_model()[0]
    x = synth.synthetic_speech(64, 110250, seed=0, kind="noise").to(DEV)
    codes = model.encode(x, 3000)
    _abi.check(lib.bvc_kprobe_enable(1))
    enc_names = ["enc.0 K2048", "enc.2", "enc.4 N64", "phi_z.0 K64", "phi_z.2", "phi_z.4", "dec.0 K2048", "dec.2",
                 "dec.4", "dec.6 N80", "phi_x.0 K80", "phi_x.2", "phi_x.4", "GRU", "side dec0h", "side W_hh h",
                 "side W_ih phi_z"]
    dec_names = enc_names[6:]

    def dump(names):
        tot = 0.0
        for k, nm in enumerate(names):
            mean, mn, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int32()
            _abi.check(lib.bvc_kprobe_read(k, k + 1, ctypes.byref(mean), ctypes.byref(mn), ctypes.byref(n)))
            print(f"   node {k:2d} {nm:14s} mean {mean.value:7.2f} us  min {mn.value:7.2f} us  n {n.value}")
            tot += mean.value
        print(f"   sum of kernel bodies per step: {tot:.1f} us")

    torch.cuda.synchronize(); t0 = time.time()
    codes = model.encode(x, 3000)
    torch.cuda.synchronize(); t1 = time.time()
    print(f"encode (probed) {1e3 * (t1 - t0):.1f} ms -> {1e6 * (t1 - t0) / 430:.1f} us per step wall")
    dump(enc_names)
    torch.cuda.synchronize(); t0 = time.time()
    model.decode(codes, 110250)
    torch.cuda.synchronize(); t1 = time.time()
    print(f"decode (probed, incl. vocoder) {1e3 * (t1 - t0):.1f} ms")
    dump(dec_names)
    _abi.check(lib.bvc_kprobe_enable(0))


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "frontend", "bvrnn", "vocoder", "facade", "timing"]
    for w in which:
        section(w, globals()[w])
