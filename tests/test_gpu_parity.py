"""Parity of the HIP path (through the C ABI) against the CPU oracle and the golden vectors captured
from the reference.  Needs the MI355X: run with ``-m gpu``.

Bars (BASELINE.json north_star): code bits exact; waveform <= 1e-4 RMS.  The mel front-end and the
intermediate float tensors carry their own tolerances, written next to each assert.
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def env():
    from gpu_common import make_model
    return make_model(True, 1024)


@pytest.fixture(scope="module")
def lib():
    from bvcodec import _abi
    return _abi.load()


# ----------------------------------------------------------------------------------- GEMM kernels
@pytest.mark.parametrize("M", [1, 7, 16, 64, 100])
@pytest.mark.parametrize("N,K", [(1024, 1024), (64, 1024), (1024, 64), (1024, 80), (80, 1024), (1024, 2048)])
@pytest.mark.parametrize("act", [0, 1])
def test_skinny_linear(lib, M, N, K, act):
    from bvcodec import _abi
    g = torch.Generator().manual_seed(M * 131 + N + K + act)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / np.sqrt(K)
    b = torch.randn(N, generator=g)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    if act:
        ref = torch.nn.functional.elu(ref)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    y = torch.full((M, N), float("nan"), device=DEV)
    _abi.check(lib.bvc_test_linear(_abi.ptr(xd), _abi.ptr(wd), _abi.ptr(bd), M, N, K, act, _abi.ptr(y),
                                   _abi.current_stream(torch.device(DEV))))
    torch.cuda.synchronize()
    err = (y.cpu().double() - ref).abs().max().item()
    assert err < 2e-5, err          # fp32 accumulation over K <= 2048 of O(1) terms


@pytest.mark.parametrize("M", [5, 128, 1000])
@pytest.mark.parametrize("N,K", [(1024, 80), (1024, 1024), (64, 64)])
def test_batched_linear(lib, M, N, K):
    from bvcodec import _abi
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / np.sqrt(K)
    b = torch.randn(N, generator=g)
    ref = torch.nn.functional.elu(torch.nn.functional.linear(x.double(), w.double(), b.double()))
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    y = torch.full((M, N), float("nan"), device=DEV)
    _abi.check(lib.bvc_test_linear_batched(_abi.ptr(xd), _abi.ptr(wd), _abi.ptr(bd), M, N, K, 1, _abi.ptr(y),
                                           _abi.current_stream(torch.device(DEV))))
    torch.cuda.synchronize()
    assert (y.cpu().double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("M", [5, 100, 300, 2000])
@pytest.mark.parametrize("N,K", [(1024, 1024), (1024, 80), (1024, 64), (80, 1024), (1024, 256), (1024, 2048), (3072, 1024)])
@pytest.mark.parametrize("act", [0, 1])
def test_recurrent_layer_kernel_and_batched_kernels_give_the_same_bits(lib, M, N, K, act):
    """One order of summation per output (head of k_gemm.hip: eight chunks of the k-blocks, each a chain from zero, added in order,
    then the bias): the launch-per-layer kernel (K split over eight waves) and the batched kernels (one accumulator chain per
    output, folded per chunk) must agree bit for bit - this is what makes a streaming hop, which runs phi_x / phi_z frame by frame
    on the former, emit the same codes as the offline call, which runs them on the latter."""
    from bvcodec import _abi
    g = torch.Generator().manual_seed(7 * M + N + K + act)
    x = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) / np.sqrt(K)).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    st = _abi.current_stream(torch.device(DEV))
    y1 = torch.full((M, N), float("nan"), device=DEV)
    y2 = torch.full((M, N), float("nan"), device=DEV)
    _abi.check(lib.bvc_test_linear(_abi.ptr(x), _abi.ptr(w), _abi.ptr(b), M, N, K, act, _abi.ptr(y1), st))
    _abi.check(lib.bvc_test_linear_batched(_abi.ptr(x), _abi.ptr(w), _abi.ptr(b), M, N, K, act, _abi.ptr(y2), st))
    torch.cuda.synchronize()
    assert torch.equal(y1, y2), (y1 - y2).abs().max().item()


@pytest.mark.parametrize("M,tail_rows", [(11129, 32), (11982, 64)])
def test_batched_linear_tail_tiles_equal_full_tiles(lib, M, tail_rows):
    """launch_gemm_batched: the rows of the last, partly filled round of 128-row tiles go out as a second launch of quarter-
    (87 row blocks: 23 left over) or half-height tiles (94: 30 left over).  Same k order per accumulator: the tail rows must have
    the bits they get when the same rows are computed alone (all of them in full-height tiles), and be right."""
    from bvcodec import _abi
    N = K = 1024
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / np.sqrt(K)
    b = torch.randn(N, generator=g)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    st = _abi.current_stream(torch.device(DEV))
    y = torch.full((M, N), float("nan"), device=DEV)
    _abi.check(lib.bvc_test_linear_batched(_abi.ptr(xd), _abi.ptr(wd), _abi.ptr(bd), M, N, K, 1, _abi.ptr(y), st))
    m_off = 64 * 128                                         # one full round: 512 slots / 8 column blocks
    xt = xd[m_off:].contiguous()
    yt = torch.full((M - m_off, N), float("nan"), device=DEV)
    _abi.check(lib.bvc_test_linear_batched(_abi.ptr(xt), _abi.ptr(wd), _abi.ptr(bd), M - m_off, N, K, 1, _abi.ptr(yt), st))
    torch.cuda.synchronize()
    assert torch.equal(y[m_off:], yt)
    rows = torch.cat([torch.arange(0, 300), torch.arange(m_off - 100, M)])
    ref = torch.nn.functional.elu(torch.nn.functional.linear(x[rows].double(), w.double(), b.double()))
    assert (y[rows].cpu().double() - ref).abs().max().item() < 2e-5


# ----------------------------------------------------------------------------------- front-end (A3)
def _mel_close(got, ref):
    """|d mel_lin| <= 1e-6 + 3e-6 * mel_lin: in quiet bands of tonal input the float32 FFT's own
    rounding noise (different in every FFT implementation) dominates, so compare linearly."""
    lg, lr = np.exp(got.astype(np.float64)), np.exp(ref.astype(np.float64))
    return bool((np.abs(lg - lr) <= 1e-6 + 3e-6 * lr).all())


@pytest.mark.parametrize("name", ["g1_mel", "g1_mel_short"])
def test_frontend_vs_reference_golden(env, name):
    model = env[0]
    g = load_golden(name)
    mel = model.mel_spectrogram(t(g["x"]).to(DEV)).cpu().numpy()          # (B,T,80)
    ref = np.transpose(g["mel"], (0, 2, 1))
    assert mel.shape == ref.shape
    assert _mel_close(mel, ref)
    if name == "g1_mel":      # broadband rows (noise, speech-like): plain log-domain tolerance
        assert np.abs(mel[0] - ref[0]).max() < 2e-5
        assert np.abs(mel[2] - ref[2]).max() < 1e-4


def test_frontend_edges_vs_reference_golden(env):
    """The front-end's floor and clamp (meldataset.py:38-39,86-90) on the reference's own outputs for digital silence, a silent stretch
    inside noise, a full-scale square wave, a clipped sine, a DC offset and one second of real speech.
    * where the reference's value IS the floor log(1e-5), ours is that same float;
    * where it is the floor or lies above 10x the floor: |d log mel| <= 2e-5 against the reference - unless the reference's own
      float32 STFT is further than that from the float64 truth of the same operator sequence (tonal input with 100 dB between peak
      and valley: clipped sine 7e-4, DC offset 2.5e-4, speech 2.3e-5), where the bar is twice the reference's own distance;
    * everywhere: the linear bound of _mel_close."""
    from oracle import frontend as ofe
    model = env[0]
    g = load_golden("g1_mel_edges")
    mel = model.mel_spectrogram(t(g["x"]).to(DEV)).cpu().numpy()
    ref = np.transpose(g["mel"], (0, 2, 1))
    assert mel.shape == ref.shape == (6, 86, 80)
    m64 = ofe.log_mel(t(g["x"]).double() * ofe.SCALING, dtype=torch.float64).permute(0, 2, 1).numpy()
    floor = np.float32(np.log(np.float32(1e-5)))
    at_floor = ref == floor
    loud = np.exp(ref.astype(np.float64)) >= 1e-4
    assert at_floor[0].all() and at_floor[1].sum() > 3000
    assert (mel[at_floor] == floor).all()                   # clamped exactly where the reference clamps, to the same float
    for i, nm in enumerate(["silence", "gap in noise", "square wave", "clipped sine", "dc offset", "real speech"]):
        sel = at_floor[i] | loud[i]
        d_ref = np.abs(mel[i].astype(np.float64) - ref[i])[sel].max()
        e_hip = np.abs(mel[i] - m64[i])[sel].max()
        e_ref = np.abs(ref[i] - m64[i])[sel].max()
        print(f"{nm}: at floor {int(at_floor[i].sum())}, loud {int(loud[i].sum())} of {ref[i].size}; there max |dlog| vs reference {d_ref:.2e}; "
              f"vs float64 truth: ours {e_hip:.2e}, the reference's {e_ref:.2e}", flush=True)
        assert d_ref <= 2e-5 or e_hip <= 2.0 * e_ref, (nm, d_ref, e_hip, e_ref)
    assert _mel_close(mel, ref)
    assert np.isfinite(mel).all()


def test_frontend_vs_float64_truth(env):
    from oracle import frontend as ofe
    from bvcodec import synth
    model = env[0]
    x = synth.synthetic_speech(4, 22050, seed=3, kind="speech")
    mel = model.mel_spectrogram(x.to(DEV)).cpu().numpy()
    m64 = ofe.log_mel(x.double() * ofe.SCALING, dtype=torch.float64).permute(0, 2, 1).numpy()
    m32 = ofe.log_mel(x * ofe.SCALING).permute(0, 2, 1).numpy()
    e_hip = np.abs(mel - m64).max()
    e_ref = np.abs(m32 - m64).max()
    assert mel.shape == (4, 86, 80)
    assert e_hip < max(2 * e_ref, 2e-5), (e_hip, e_ref)     # as close to the truth as the fp32 reference path is


# ----------------------------------------------------------------------------------- BVRNN (A5-A7)
@pytest.mark.parametrize("tag,h_dim,var_bit", [("h1024_var", 1024, True), ("h1024_fix", 1024, False),
                                              ("h64_var", 64, True)])
def test_bvrnn_vs_reference_golden(tag, h_dim, var_bit):
    from gpu_common import make_model
    model = make_model(var_bit, h_dim)[0]
    g = load_golden(f"g3_bvrnn_{tag}")
    B = g["y"].shape[0]
    h0 = torch.zeros(1, B, h_dim, device=DEV)
    codes, all_h, prob = model.bvrnn.encode(t(g["y"]).to(DEV), t(g["bits"]).to(DEV), h0, return_prob=True)
    assert np.abs(prob.cpu().numpy() - g["prob"]).max() < 2e-6
    assert np.array_equal(codes.cpu().numpy(), g["codes"])                  # bit-exact codes
    assert np.abs(all_h.cpu().numpy() - g["all_h"]).max() < 5e-6
    mel, hT = model.bvrnn.decode(t(g["codes"]).to(DEV), h0)
    assert np.abs(mel.cpu().numpy() - g["mel_hat"]).max() < 5e-5
    assert np.abs(hT[0].cpu().numpy() - g["h_T"]).max() < 5e-6


@pytest.mark.parametrize("tag,h_dim,var_bit", [("h1024_var", 1024, True), ("h1024_fix", 1024, False),
                                              ("h64_var", 64, True)])
def test_bvrnn_forward_vs_reference_golden(tag, h_dim, var_bit):
    """bvc_bvrnn_forward == the reference's BVRNN.forward (bvrnn.py:86-160) on the recorded random numbers:
    greedy / sampled, p_use_gen in {0, 0.5, 1}."""
    from gpu_common import make_model
    model = make_model(var_bit, h_dim)[0]
    g = load_golden(f"g8_bvrnn_forward_{tag}")
    for mode in range(4):
        k = f"m{mode}_"
        greedy = bool(g[k + "greedy"])
        noise = None if greedy else t(g[k + "noise"])
        dec, kld, ex = model.bvrnn(t(g[k + "y"]).to(DEV), float(g[k + "p_use_gen"]), greedy, t(g[k + "bits"]).to(DEV),
                                   r=t(g[k + "r"]), noise=noise, return_all=True)
        assert np.abs(ex["prob"].cpu().numpy() - g[k + "prob"]).max() < 3e-6, mode
        assert np.abs(ex["prior"].cpu().numpy() - g[k + "prior"]).max() < 3e-6, mode
        assert np.abs(dec.cpu().numpy() - g[k + "dec"]).max() < 5e-5, mode
        assert abs(float(kld) - float(g[k + "kld"])) < 2e-6 * max(1.0, abs(float(g[k + "kld"]))), mode
    # default randomness: same numbers as the reference draws after the same torch seed (global CPU generator)
    torch.manual_seed(int(g["m2_torch_seed"]))
    dec2, kld2 = model.bvrnn(t(g["m2_y"]).to(DEV), float(g["m2_p_use_gen"]), False, t(g["m2_bits"]).to(DEV))
    assert np.abs(dec2.cpu().numpy() - g["m2_dec"]).max() < 5e-5
    assert abs(float(kld2) - float(g["m2_kld"])) < 2e-6


def test_bvrnn_forward_vs_oracle_larger(env):
    """B=20 (ragged row tile), T=40, sampled with mixed teacher forcing: sample bits equal the oracle's except at
    ties of the rounding, decoder output and KLD agree."""
    from oracle import bvrnn as obv
    model, conf, vr, _ = env
    rng = np.random.default_rng(9)
    B, T = 20, 40
    y = torch.from_numpy((-4.0 + 1.6 * rng.standard_normal((B, T, 80))).astype(np.float32))
    bits = torch.from_numpy(rng.integers(0, 65, size=(B, T)).astype(np.float32))
    gen = torch.Generator().manual_seed(77)
    r, noise = obv.draw_randomness(T, B, 64, False, generator=gen)
    o = obv.forward(vr, y, 0.3, False, bits, r, noise)
    dec, kld, ex = model.bvrnn(y.to(DEV), 0.3, False, bits.to(DEV), r=r, noise=noise, return_all=True)
    z, zo = ex["z"].cpu(), o["z"]
    diff = (z - zo).abs() > 0.25
    first_bad = T
    for b in range(B):
        if diff[b].any():
            t0 = int(diff[b].any(dim=1).nonzero()[0])
            bad = diff[b, t0].nonzero().flatten()
            assert ((o["arg"][b, t0, bad] - 0.5).abs() < 1e-5).all(), (b, t0)     # only ties may flip
            first_bad = min(first_bad, t0)
    # frames before the first flipped tie (normally: all of them) agree closely
    assert first_bad > 0
    assert (dec.cpu()[:, :first_bad] - o["dec"][:, :first_bad]).abs().max() < 1e-4
    assert (ex["kld_frames"].cpu()[:first_bad] - o["kld_frames"][:first_bad]).abs().max() < 1e-5
    if first_bad == T:
        assert abs(float(kld) - float(o["kld"])) < 1e-5
    # argument checking: a frame conditioned on a state that is never updated is rejected
    with pytest.raises(RuntimeError):
        model.bvrnn(y.to(DEV), 0.0, True, bits.to(DEV), r=torch.full((T,), -1.0))


def test_bvrnn_free_running_vs_oracle(env):
    """Larger free-running case: any differing bit must be a tie of the reference arithmetic
    (|p-0.5| < 1e-5 in the oracle) at the FIRST differing frame of that utterance."""
    from oracle import bvrnn as obv
    model, conf, vr, _ = env
    rng = np.random.default_rng(5)
    B, T = 6, 48
    y = torch.from_numpy((-4.0 + 1.6 * rng.standard_normal((B, T, 80))).astype(np.float32))
    bits = torch.from_numpy(rng.integers(10, 65, size=(B, T)).astype(np.float32))
    r = obv.encode(vr, y, bits, torch.zeros(B, 1024))
    codes, all_h = model.bvrnn.encode(y.to(DEV), bits.to(DEV), torch.zeros(1, B, 1024, device=DEV))
    codes = codes.cpu()
    diff = (codes != r["codes"])
    for b in range(B):
        if diff[b].any():
            t0 = int(diff[b].any(dim=1).nonzero()[0])
            bad = diff[b, t0].nonzero().flatten()
            assert ((r["prob"][b, t0, bad] - 0.5).abs() < 1e-5).all(), (b, t0)
    # teacher-forced: restart every frame from the ORACLE's state -> every non-tie bit agrees
    # (run per frame through the same kernels: T independent one-frame encodes)
    mism = 0
    for tt in range(0, T, 7):
        c1, _ = model.bvrnn.encode(y[:, tt:tt + 1].to(DEV), bits[:, tt:tt + 1].to(DEV),
                                   r["all_h"][:, tt].unsqueeze(0).to(DEV))
        d = (c1.cpu()[:, 0] != r["codes"][:, tt])
        assert ((r["prob"][:, tt][d] - 0.5).abs() < 1e-5).all()
        mism += int(d.sum())
    assert mism <= 2


def test_bvrnn_chunked_decode_is_exact(env):
    """BVRNN.decode carried over chunks (state hand-over) is bit-identical to one call."""
    model = env[0]
    rng = np.random.default_rng(9)
    z = torch.from_numpy(rng.integers(0, 2, size=(3, 40, 64)).astype(np.float32)).to(DEV)
    h0 = torch.zeros(1, 3, 1024, device=DEV)
    full, hT = model.bvrnn.decode(z, h0)
    a, ha = model.bvrnn.decode(z[:, :17], h0)
    b, hb = model.bvrnn.decode(z[:, 17:], ha)
    assert torch.equal(torch.cat([a, b], 1), full) and torch.equal(hb, hT)


# ----------------------------------------------------------------------------------- BigVGAN (A8)
def test_vocoder_taps_vs_reference_golden(env, lib):
    from bvcodec import _abi
    model = env[0]
    from gpu_common import make_model
    g = load_golden("g5_bigvgan_taps")
    # the taps fixture uses generator seed 1235 == make_model's seed+1
    mel = t(g["mel"]).permute(0, 2, 1).contiguous().to(DEV)                 # (1,T,80)
    B, T = mel.shape[0], mel.shape[1]
    eng = model.engine(mel)
    ws, nws = eng.workspace(B, T)
    names = ["conv_pre"]
    for i in range(4):
        names += [f"up{i}", f"stage{i}"]
    for which, nm in enumerate(names):
        n = ctypes.c_int64()
        _abi.check(lib.bvc_test_vocoder_tap(eng.handle, _abi.ptr(mel), B, T, which, None, ctypes.byref(n), ws, nws,
                                            eng.stream()))
        out = torch.full((B, n.value), float("nan"), device=DEV)
        _abi.check(lib.bvc_test_vocoder_tap(eng.handle, _abi.ptr(mel), B, T, which, _abi.ptr(out), ctypes.byref(n),
                                            ws, nws, eng.stream()))
        torch.cuda.synchronize()
        if nm.startswith("stage"):
            i = int(nm[5:])
            ref = (g[f"res{i}_0"] + g[f"res{i}_1"] + g[f"res{i}_2"]) / 3
        else:
            ref = g[nm]
        C = ref.shape[1]
        got = out.cpu().numpy().reshape(B, -1, C).transpose(0, 2, 1)          # -> (B,C,L)
        assert got.shape == ref.shape, (nm, got.shape, ref.shape)
        err = np.abs(got - ref).max()
        scale = max(1.0, np.abs(ref).max())
        assert err < 2e-5 * scale, (nm, err, scale)


def test_vocoder_vs_reference_golden(env):
    model = env[0]
    g = load_golden("g5_bigvgan")
    mel = t(g["mel"]).to(DEV)                                                 # (B,80,T) like the reference
    T = mel.shape[2]
    for length in (8192, 8392, 10 ** 9):
        w = model.vocoder(mel, length).cpu().numpy()
        ref = g[f"wav_{length}"]
        assert w.shape == ref.shape and w.shape[2] == min(length, 256 * T + 294)
        assert np.sqrt(((w - ref) ** 2).mean()) < 1e-5
        assert np.abs(w - ref).max() < 1e-4


# ----------------------------------------------------------------------------------- facade (A2, A9)
@pytest.mark.parametrize("tag", ["var", "fix"])
def test_facade_vs_reference_golden(tag):
    from gpu_common import make_model
    model = make_model(tag == "var", 1024)[0]
    g = load_golden(f"g6_e2e_{tag}")
    x = t(g["x"]).to(DEV)
    for br in ((3000, 1500, 6000) if tag == "var" else (3000,)):
        codes = model.encode(x, br)
        assert codes.shape == g[f"codes_{br}"].shape and codes.dtype == torch.float32
        assert np.array_equal(codes.cpu().numpy(), g[f"codes_{br}"]), br       # bit-exact codes
        wav = model.decode(t(g[f"codes_{br}"]).to(DEV), x.shape[1])
        ref = g[f"wav_{br}"]
        assert wav.shape == ref.shape
        assert np.sqrt(((wav.cpu().numpy() - ref) ** 2).mean()) < 1e-4         # north_star waveform bar
    full = model(x, 3000)
    assert np.sqrt(((full.cpu().numpy() - g["wav_3000"]) ** 2).mean()) < 1e-4
    un = model.decode(t(g["codes_3000"]).to(DEV), 10 ** 9)
    assert un.shape == g["wav_untrimmed_3000"].shape
    assert np.sqrt(((un.cpu().numpy() - g["wav_untrimmed_3000"]) ** 2).mean()) < 1e-4


def test_facade_edges_vs_reference_golden(env):
    """encode / decode at 3000 bit/s on the edge inputs against the reference's outputs (tests/golden/make_golden.py --sets edges).
    No tie-margin seed search went into this fixture, so the tie rule applies: a bit may differ only where the reference's own
    probability is within 1e-5 of 0.5, judged at the first differing frame of an utterance; waveform RMS <= 1e-4."""
    from parity_stats import divergence_stats
    model = env[0]
    g = load_golden("g6_e2e_edges")
    x = t(g["x"]).to(DEV)
    codes = model.encode(x, 3000).cpu().numpy()
    st = divergence_stats(codes, g["codes_3000"], g["prob_3000"], active_bits=35)
    print({k: st[k] for k in ("diverged_utterances", "first_divergent_frames", "max_first_divergence_margin", "bits_within_1e-5_of_a_tie")}, flush=True)
    assert st["max_first_divergence_margin"] < 1e-5, st
    mel = model.mel_spectrogram(x)
    bits = torch.full(mel.shape[:2], 35.0, device=DEV)
    _, _, prob = model.bvrnn.encode(mel, bits, torch.zeros(1, x.shape[0], 1024, device=DEV), return_prob=True)
    ok = [b for b in range(x.shape[0]) if not (codes[b] != g["codes_3000"][b]).any()]
    assert len(ok) >= 5                                       # (all six in practice)
    assert np.abs(prob.cpu().numpy()[ok] - g["prob_3000"][ok]).max() < 2e-6
    wav = model.decode(t(g["codes_3000"]).to(DEV), x.shape[1]).cpu().numpy()
    assert wav.shape == g["wav_3000"].shape and np.isfinite(wav).all()
    rms = np.sqrt(((wav - g["wav_3000"]) ** 2).mean(axis=1))
    print("waveform rms error per input", rms, flush=True)
    assert rms.max() < 1e-4                                   # north_star waveform bar, every input on its own


def test_facade_accepts_cpu_tensors_and_returns_on_caller_device(env):
    model = env[0]
    g = load_golden("g6_e2e_var")
    x = t(g["x"])[:1]
    codes = model.encode(x, 3000)
    assert codes.device.type == "cpu" and np.array_equal(codes.numpy(), g["codes_3000"][:1])


def test_facade_error_behaviour(env):
    model = env[0]
    with pytest.raises(RuntimeError):
        model.encode(torch.zeros(1, 512, device=DEV), 3000)      # reflect pad needs L > 512
    with pytest.raises(RuntimeError):
        model.decode(torch.zeros(1, 4, 63, device=DEV), 1000)


# ----------------------------------------------------------------------------------- properties at size
def test_batch_invariance_and_determinism(env):
    """An utterance's codes and waveform do not depend on what else is in the batch or on the run."""
    from bvcodec import synth
    model = env[0]
    x = synth.synthetic_speech(24, 256 * 40 + 13, seed=7, kind="speech").to(DEV)
    c_all = model.encode(x, 3000)
    c_again = model.encode(x, 3000)
    assert torch.equal(c_all, c_again)
    c_sub = model.encode(x[5:8], 3000)
    assert torch.equal(c_all[5:8], c_sub)
    w_all = model.decode(c_all, x.shape[1])
    w_sub = model.decode(c_all[5:8], x.shape[1])
    assert torch.equal(w_all[5:8], w_sub)
    assert torch.isfinite(w_all).all()


def test_vocoder_is_causal(env):
    """First 256*T' samples depend only on the first T' frames."""
    model = env[0]
    rng = np.random.default_rng(3)
    mel = torch.from_numpy((-4 + 1.6 * rng.standard_normal((2, 80, 24))).astype(np.float32)).to(DEV)
    full = model.vocoder(mel, 10 ** 9)
    part = model.vocoder(mel[:, :, :10].contiguous(), 10 ** 9)
    assert torch.equal(full[:, :, :2560], part[:, :, :2560])


@pytest.mark.parametrize("option", ["vocoder_full_tiles", "vocoder_c16_kernel"])
def test_vocoder_full_tile_c8_kernel_equals_generic(env, option):
    """The C = 8 stage runs on amp_pair8_kernel (two output rows per MFMA tile, k_vocoder.hip; the generic kernel pads the
    eight channels to sixteen columns), the C = 16 stage on amp_pair16_kernel (persistent, operands swapped, 16-byte
    epilogues).  Every output accumulates the same products in the same order as in the generic kernel: identical bits, for
    lengths that end inside / on / beyond a tile and for more rows than one workgroup tile per utterance."""
    model = env[0]
    eng = model.engine(torch.empty(0, device=DEV))
    rng = np.random.default_rng(11)
    try:
        for B, T, length in ((3, 9, 10 ** 9), (2, 40, 256 * 40 + 100), (1, 3, 700), (5, 1, 10 ** 9), (70, 33, 256 * 33 - 9)):
            mel = torch.from_numpy((-4 + 1.6 * rng.standard_normal((B, 80, T))).astype(np.float32)).to(DEV)
            eng.set_option(option, 1)
            a = model.vocoder(mel, length)
            eng.set_option(option, 0)
            b = model.vocoder(mel, length)
            assert a.shape == b.shape and torch.isfinite(a).all()
            assert torch.equal(a, b), (option, B, T, length, (a - b).abs().max().item())
    finally:
        eng.set_option(option, 1)


def test_full_size_config_roundtrip_properties():
    """BASELINE configs[1] shape (64 x 5 s @ 3 kbit/s) through encode+decode: shapes, value sets,
    finiteness, and agreement of a sampled utterance with the oracle."""
    from gpu_common import make_model
    from bvcodec import synth
    from oracle import codec as ocodec
    model, conf, vr, ge = make_model(True, 1024)
    x = synth.synthetic_speech(64, 110250, seed=0, kind="noise").to(DEV)
    codes = model.encode(x, 3000)
    assert codes.shape == (64, 430, 64)
    vals = torch.unique(codes).cpu().tolist()
    assert set(vals) <= {0.0, 0.5, 1.0}
    assert (codes[:, :, 35:] == 0.5).all() and (codes[:, :, :35] != 0.5).all()
    wav = model.decode(codes, x.shape[1])
    assert wav.shape == (64, 110250) and torch.isfinite(wav).all()
    # one utterance, first second, against the oracle (free-running; ties excluded as above)
    oc = ocodec.OracleCodec(conf, vr, ge)
    r = oc.encode(x[3:4, :22050].cpu(), 3000, full=True)
    got = model.encode(x[3:4, :22050], 3000).cpu()
    diff = got != r["codes"]
    if diff.any():
        t0 = int(diff[0].any(dim=1).nonzero()[0])
        assert ((r["prob"][0, t0][diff[0, t0]] - 0.5).abs() < 1e-5).all()
    ref_wav = oc.decode(got, 22050)
    w = model.decode(got.to(DEV), 22050).cpu()
    assert float((w - ref_wav).pow(2).mean().sqrt()) < 1e-4


def test_config3_per_gpu_shard_10s_three_bitrates():
    """BASELINE configs[3]: the 64-utterance per-GPU shard of the 512 x 10 s job at 1500 / 3000 / 6000 bit/s, checked
    through size-independent properties: bit-mask structure per bitrate (17 / 35 / 64 active bits), causality (the
    first 5 s coded alone give the same codes for every frame that does not touch the right reflect padding), the
    decoder's prefix property, and the bit-packed wire size."""
    from gpu_common import make_model
    from bvcodec import synth
    model = make_model(True, 1024)[0]
    x = synth.synthetic_speech(64, 220500, seed=2, kind="noise").to(DEV)
    half = 110250
    for bitrate, nbits in ((1500, 17), (3000, 35), (6000, 64)):
        codes = model.encode(x, bitrate)
        assert codes.shape == (64, 861, 64)
        assert (codes[:, :, nbits:] == 0.5).all() and (codes[:, :, :nbits] != 0.5).all()
        assert model.active_bits(bitrate) == nbits
        first = model.encode(x[:, :half], bitrate)                  # 430 frames; the last 2 see reflected samples
        assert torch.equal(first[:, :428], codes[:, :428])
        wav = model.decode(codes, 220500)
        assert wav.shape == (64, 220500) and torch.isfinite(wav).all()
        wav_half = model.decode(codes[:, :430].contiguous(), half)   # causal decoder: a prefix of the codes gives a prefix of the waveform
        n = 256 * 430                                                 # samples past the last whole frame hold partial sums of later frames
        assert (wav_half[:, :n] - wav[:, :n]).abs().max().item() <= 1e-6
        packed = model.pack(codes, bitrate)
        assert packed.shape == (64, 861, (nbits + 7) // 8) and torch.equal(model.unpack(packed, bitrate), codes)


# ----------------------------------------------------------------------------------- BASELINE configs / edge cases
def test_config0_single_10s_utterance_fixed_64bit():
    """BASELINE configs[0]: one 10 s utterance, config_64bit (bitrate ignored), against the oracle."""
    from gpu_common import make_model
    from bvcodec import synth
    from oracle import codec as ocodec
    model, conf, vr, ge = make_model(False, 1024)
    x = synth.synthetic_speech(1, 220500, seed=31, kind="speech")
    codes = model.encode(x.to(DEV), 1234).cpu()
    assert codes.shape == (1, 861, 64) and set(torch.unique(codes).tolist()) <= {0.0, 1.0}
    oc = ocodec.OracleCodec(conf, vr, ge)
    r = oc.encode(x, 1234, full=True)
    diff = codes != r["codes"]
    if diff.any():                                   # only a tie of the reference arithmetic may differ
        t0 = int(diff[0].any(dim=1).nonzero()[0])
        assert ((r["prob"][0, t0][diff[0, t0]] - 0.5).abs() < 1e-5).all()
    wav = model.decode(codes.to(DEV), 220500).cpu()
    ref = oc.decode(codes, 220500)
    assert wav.shape == (1, 220500)
    assert float((wav - ref).pow(2).mean().sqrt()) < 1e-4


@pytest.mark.parametrize("B,L", [(1, 513), (1, 767), (5, 768), (17, 1030), (33, 2049)])
def test_short_and_ragged_shapes_vs_oracle(env, B, L):
    """Shortest legal inputs (reflect pad needs L > 512), batch sizes that are not multiples of the
    16-row MFMA tile, lengths that are not multiples of the hop."""
    from bvcodec import synth
    from oracle import codec as ocodec
    model, conf, vr, ge = env
    x = synth.synthetic_speech(B, L, seed=B + L, kind="noise")
    codes = model.encode(x.to(DEV), 3000).cpu()
    T = L // 256
    assert codes.shape == (B, T, 64)
    oc = ocodec.OracleCodec(conf, vr, ge)
    r = oc.encode(x, 3000, full=True)
    diff = codes != r["codes"]
    assert not bool((diff & ((r["prob"] - 0.5).abs() > 1e-5)).any())
    wav = model.decode(r["codes"].to(DEV), L).cpu()
    ref = oc.decode(r["codes"], L)
    assert wav.shape == ref.shape == (B, min(L, 256 * T + 294))
    assert float((wav - ref).pow(2).mean().sqrt()) < 1e-5


def test_bitrate_extremes(env):
    model = env[0]
    from bvcodec import synth
    x = synth.synthetic_speech(2, 256 * 12, seed=2, kind="speech").to(DEV)
    c0 = model.encode(x, 0)
    assert (c0 == 0.5).all()                                          # 0 active bits: every position masked
    c_hi = model.encode(x, 10 ** 6)
    assert (c_hi != 0.5).all()                                        # saturates at z_dim bits
    assert torch.equal(c_hi, model.encode(x, 5512.5))
    w = model.decode(c0, 256 * 12)
    assert torch.isfinite(w).all()


def test_nonzero_initial_state_and_returned_state(env):
    """BVRNN.decode(z, h) with a non-zero h (reference signature, bvrnn.py:211) against the oracle."""
    from oracle import bvrnn as obv
    model, conf, vr, _ = env
    rng = np.random.default_rng(12)
    z = torch.from_numpy(rng.integers(0, 2, size=(4, 9, 64)).astype(np.float32))
    h0 = torch.from_numpy((0.3 * rng.standard_normal((4, 1024))).astype(np.float32))
    mel, hT = model.bvrnn.decode(z.to(DEV), h0.unsqueeze(0).to(DEV))
    r = obv.decode(vr, z, h0)
    assert np.abs(mel.cpu().numpy() - r["mel"].numpy()).max() < 5e-5
    assert np.abs(hT[0].cpu().numpy() - r["h_last"].numpy()).max() < 5e-6


def test_initial_state_at_any_alignment(env):
    """The persistent launch takes its initial state, the sentinel fill and its arguments from ONE preparation kernel when the state is
    16-byte aligned, from three separate kernels when it is not (a C caller may pass any float pointer): same bits, and the state after
    the last frame continues a split sequence exactly."""
    model = env[0]
    rng = np.random.default_rng(5)
    B, T = 7, 11
    z = torch.from_numpy(rng.integers(0, 2, size=(B, T, 64)).astype(np.float32)).to(DEV)
    h0 = torch.from_numpy((0.3 * rng.standard_normal((B, 1024))).astype(np.float32)).to(DEV)
    buf = torch.empty(B * 1024 + 1, device=DEV)
    h0_odd = buf[1:].view(B, 1024)
    h0_odd.copy_(h0)
    assert h0.data_ptr() % 16 == 0 and h0_odd.data_ptr() % 16 == 4 and h0_odd.is_contiguous()
    mel_a, hT_a = model.bvrnn.decode(z, h0.unsqueeze(0))
    mel_b, hT_b = model.bvrnn.decode(z, h0_odd.unsqueeze(0))
    assert torch.equal(mel_a, mel_b) and torch.equal(hT_a, hT_b)
    mel_1, h_mid = model.bvrnn.decode(z[:, :4].contiguous(), h0.unsqueeze(0))
    mel_2, hT_c = model.bvrnn.decode(z[:, 4:].contiguous(), h_mid)
    assert torch.equal(torch.cat([mel_1, mel_2], 1), mel_a) and torch.equal(hT_c, hT_a)


def test_execution_variants_agree(env):
    """The hipGraph-replayed default, the eager fallback (BVC_NO_GRAPH=1, what the library falls back to
    when stream capture is unavailable) and the unfused vocoder (BVC_UNFUSED_AMP=1) run the same
    arithmetic: identical codes and mel; the opt-in side-branch schedule (BVC_SIDE_BRANCH=1) splits two
    dot products, so it agrees to rounding."""
    from gpu_common import make_model
    from bvcodec import synth
    base = env[0]
    x = synth.synthetic_speech(5, 256 * 21 + 9, seed=77, kind="speech").to(DEV)
    codes = base.encode(x, 3000)
    wav = base.decode(codes, x.shape[1])
    eager = make_model(True, 1024, env={"BVC_NO_GRAPH": "1"})[0]
    assert torch.equal(eager.encode(x, 3000), codes)
    assert torch.equal(eager.decode(codes, x.shape[1]), wav)
    unfused = make_model(True, 1024, env={"BVC_UNFUSED_AMP": "1"})[0]
    assert (unfused.decode(codes, x.shape[1]) - wav).abs().max().item() < 2e-6
    side = make_model(True, 1024, env={"BVC_SIDE_BRANCH": "1"})[0]
    mel_a, _ = base.bvrnn.decode(codes, torch.zeros(1, 5, 1024, device=DEV))
    mel_b, _ = side.bvrnn.decode(codes, torch.zeros(1, 5, 1024, device=DEV))
    assert (mel_a - mel_b).abs().max().item() < 1e-5
    # default decode batches the phi_z halves of dec.0 / the GRU input product over all frames; BVC_NO_PRECOMP=1 keeps
    # them inside the recurrence: one more rounding per split dot product
    inrec = make_model(True, 1024, env={"BVC_NO_PRECOMP": "1"})[0]
    mel_c, h_c = inrec.bvrnn.decode(codes, torch.zeros(1, 5, 1024, device=DEV))
    assert (mel_a - mel_c).abs().max().item() < 1e-5
    assert torch.equal(inrec.encode(x, 3000), codes)


@pytest.mark.parametrize("B", [5, 64, 130])
def test_fused_forward_equals_encode_plus_decode_to_rounding(env, B):
    """model.forward_fused (bvc_forward): the decoder outputs of the ENCODER's frame loop (bvrnn.py:198-204) go to the vocoder, no second
    recurrence.  Codes are encode()'s bit for bit; the waveform agrees with forward() = decode(encode(x)) (bvrnn_codec_model.py:73-76) to
    rounding (the halves of dec.0 and of the GRU's input gates are summed in another order) and with the oracle's forward within the
    north-star bar; the same bits on the persistent kernel (one group per workgroup / interleaved chains) and on the layer schedule, with and
    without the folded hop."""
    from bvcodec import synth
    from oracle import codec as ocodec
    model, conf, vr, ge = env
    eng = model.engine()
    L = 256 * 30 + 100
    x = synth.synthetic_speech(B, L, seed=40 + B, kind="speech").to(DEV)
    codes = model.encode(x, 3000)
    ref = model.forward(x, 3000)
    out = {}
    try:
        for sched in ("persistent", "layers"):
            model.set_recurrence(sched)
            for fold in (1, 0):
                eng.set_option("encode_fold", fold)
                c, w = model.forward_fused(x, 3000, return_codes=True)
                assert torch.equal(c, codes), (sched, fold)
                out[(sched, fold)] = w
        torch.cuda.synchronize()
        model.check_status()
    finally:
        eng.set_option("encode_fold", 1)
        model.set_recurrence("auto")
    for fold in (1, 0):
        assert torch.equal(out[("persistent", fold)], out[("layers", fold)]), fold
    for k, w in out.items():
        assert w.shape == ref.shape
        assert float((w - ref).pow(2).mean().sqrt()) < 1e-5, k
    oc = ocodec.OracleCodec(conf, vr, ge)
    pick = [0, B - 1]
    ow = oc.forward(x[pick].cpu(), 3000)
    assert float((out[("persistent", 1)][pick].cpu() - ow).pow(2).mean().sqrt()) < 1e-4
