#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--ref /root/reference]

The reference checkout is imported read-only from ``--ref``; nothing of it is copied.  Three of its
imports are absent from the image and not installable offline: ``toml`` (the facade's
config reader, bvrnn_codec_model.py:8), ``librosa.filters.mel`` (meldataset.py:15) and
``librosa.util.normalize`` (meldataset.py:13, never called on the path).  They are provided as
in-process stand-ins: ``toml.load`` -> ``tomli``; ``librosa.filters.mel`` -> the oracle's restatement
of the published Slaney construction (oracle/melbank.py; pinned independently against
``transformers.audio_utils.mel_filter_bank`` in tests/test_oracle_golden.py).  Every other
operation is the reference's own code running on PyTorch-CPU.

Weights are seeded synthetic checkpoints in the reference's checkpoint format
(bvcodec/synth.py); only inputs and expected outputs are stored, weights are regenerated from
the seed by the tests.
"""
import argparse
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from bvcodec import config as bconfig, synth          # noqa: E402  (product-side helpers: config + synthetic weights)
from oracle import melbank as omel                    # noqa: E402  (only for the librosa stand-in)


def install_standins():
    import tomli
    toml = types.ModuleType("toml")
    toml.load = lambda p: tomli.load(open(p, "rb"))
    sys.modules["toml"] = toml
    librosa = types.ModuleType("librosa")
    util = types.ModuleType("librosa.util")
    util.normalize = lambda x, *a, **k: x
    filters = types.ModuleType("librosa.filters")
    filters.mel = lambda sr, n_fft, n_mels, fmin, fmax: omel.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    librosa.util, librosa.filters = util, filters
    sys.modules.update({"librosa": librosa, "librosa.util": util, "librosa.filters": filters})


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.0f} KiB)")


def margin(prob, nbits):
    """min |p - 0.5| over ACTIVE bits (bit index < bits_per_frame)."""
    idx = torch.arange(prob.shape[-1])[None, None, :]
    act = idx < nbits[:, :, None]
    return float((prob - 0.5).abs()[act].min())


def edge_inputs(ref):
    """(6, 22050) waveforms that drive the front-end's edge behaviour (meldataset.py:38-39,86-90: sqrt(.+1e-9), clamp(1e-5), log):
    digital silence, a silent stretch inside an utterance, full-scale square wave, clipped sine, DC offset, and one second
    of real speech from the reference's listening-test material, prepared as example.py:12-17 does."""
    import scipy.io.wavfile
    import scipy.signal
    L = 22050
    n = np.arange(L, dtype=np.float64)
    x = np.zeros((6, L), dtype=np.float32)
    # 0: all-zero utterance
    noise = synth.synthetic_speech(1, L, seed=31, kind="noise")[0].numpy()
    x[1] = noise
    x[1, 6000:6000 + 11025] = 0.0                                                    # 1: 0.5 s of zeros inside noise
    x[2] = np.where(np.sin(2 * np.pi * 220.0 * n / 22050.0) >= 0, 1.0, -1.0)        # 2: +-1.0 square wave
    x[3] = np.clip(3.0 * np.sin(2 * np.pi * 440.0 * n / 22050.0), -1.0, 1.0)        # 3: clipped sine
    x[4] = 0.5 + 0.01 * synth.synthetic_speech(1, L, seed=32, kind="noise")[0].numpy()   # 4: DC offset
    fs, d = scipy.io.wavfile.read(os.path.join(ref, "mushra_results_dataset", "audio", "stim_01", "ref.wav"))
    speech = d[:, 0].astype(np.float64) / 32768.0                                    # first channel (example.py:13)
    speech = scipy.signal.resample_poly(speech, 22050, fs)                           # example.py:16 (= 147 / 160)
    speech = speech / np.max(np.abs(speech))                                         # example.py:17
    x[5] = speech[8000:8000 + L].astype(np.float32)                                  # 5: one second of real speech
    return torch.from_numpy(x)


def edges(ref, conf, ref_cfg_var, tmp, mel_spectrogram, BVRNNCodecModel, SCALING):
    print("G1/G6 edges")
    x = edge_inputs(ref)
    mel = mel_spectrogram(x * SCALING, n_fft=1024, num_mels=80, sampling_rate=22050, hop_size=256,
                          win_size=1024, fmin=0, fmax=8000, padding_left=256)
    floor = float(np.log(1e-5))
    print("   frames at the floor per input:", [(int((mel[i] == floor).sum()), int(mel[i].numel())) for i in range(x.shape[0])])
    save("g1_mel_edges", x=x, mel=mel, scaling=np.float64(SCALING))
    p1, p2 = synth.write_checkpoints(conf, tmp, seed=1234, prefix="edges")
    model = BVRNNCodecModel(ref_cfg_var, p1, p2)
    model.eval()
    probs = []
    with torch.no_grad():
        hook = model.bvrnn.enc[5].register_forward_hook(lambda m, i, o: probs.append(o.detach().clone()))
        codes = model.encode(x, 3000)
        hook.remove()
        prob = torch.stack(probs).permute(1, 0, 2)
        wav = model.decode(codes, x.shape[1])
    print(f"   min |p-0.5| over active bits = {margin(prob, torch.full(prob.shape[:2], 35.0)):.3e} (no seed search: the tests apply the tie rule)")
    save("g6_e2e_edges", x=x, codes_3000=codes, prob_3000=prob, wav_3000=wav, seed=np.int64(1234))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--sets", default="all", help="all | base (G1-G6 as before) | edges (g1_mel_edges, g6_e2e_edges only)")
    a = ap.parse_args()
    sys.dont_write_bytecode = True
    install_standins()
    sys.path.insert(0, a.ref)
    torch.manual_seed(0)
    torch.set_num_threads(8)

    from bvrnn import BVRNN                                              # reference
    from bvrnn_codec_model import BVRNNCodecModel, SCALING               # reference
    from third_party.BigVGAN.meldataset import mel_spectrogram          # reference
    from third_party.BigVGAN.models import BigVGAN                      # reference
    from third_party.BigVGAN.env import AttrDict                        # reference

    ref_cfg_var = os.path.join(a.ref, "configs", "config_varBitRate.toml")
    ref_cfg_64 = os.path.join(a.ref, "configs", "config_64bit.toml")
    conf = bconfig.load_config(ref_cfg_var)
    conf64 = bconfig.load_config(ref_cfg_64)
    tmp = tempfile.mkdtemp(prefix="bvc_golden_")
    if a.sets in ("all", "edges"):
        edges(a.ref, conf, ref_cfg_var, tmp, mel_spectrogram, BVRNNCodecModel, SCALING)
    if a.sets == "edges":
        return

    # ---------------- G1: mel front-end -------------------------------------------------
    print("G1 mel_spectrogram")
    L = 8192 + 300
    x = synth.synthetic_speech(3, L, seed=11, kind="noise")
    n = torch.arange(L, dtype=torch.float64)
    x[1] = (0.8 * torch.sin(2 * np.pi * (50.0 + 5000.0 * n / L) * n / 22050.0)).float()   # sweep
    x[2] = synth.synthetic_speech(1, L, seed=12, kind="speech")[0]
    mel = mel_spectrogram(x * SCALING, n_fft=1024, num_mels=80, sampling_rate=22050, hop_size=256,
                          win_size=1024, fmin=0, fmax=8000, padding_left=256)
    save("g1_mel", x=x, mel=mel, scaling=np.float64(SCALING))
    # shortest legal input (reflect pad needs L > 512) and exact multiple of the hop
    xs = synth.synthetic_speech(2, 768, seed=13, kind="noise")
    mels = mel_spectrogram(xs * SCALING, 1024, 80, 22050, 256, 1024, 0, 8000, 256)
    save("g1_mel_short", x=xs, mel=mels)

    # ---------------- G3/G4: BVRNN encode / decode -------------------------------------
    for h_dim in (1024, 64):
        for var_bit in (True, False):
            if h_dim == 64 and not var_bit:
                continue
            c = dict(conf); c["h_dim"] = h_dim; c["var_bit"] = var_bit
            tag = f"h{h_dim}_{'var' if var_bit else 'fix'}"
            print(f"G3/G4 BVRNN {tag}")
            sd = synth.bvrnn_state_dict(c, seed=1234)
            net = BVRNN(80, h_dim, 64, [np.zeros(80), np.ones(80)], c["log_sigma_init"], variableBit=var_bit)
            net.load_state_dict(sd)
            net.eval()
            B, T = 2, 32
            nb_thr = 2e-5 if var_bit else 1e-5   # every ACTIVE bit is at least this far from a tie
            for in_seed in range(77, 377):
                rng = np.random.default_rng(in_seed)
                y = torch.from_numpy((-4.0 + 1.6 * rng.standard_normal((B, T, 80))).astype(np.float32))
                bits = torch.full((B, T), 35.0)
                bits[1] = torch.from_numpy(rng.integers(0, 65, size=T).astype(np.float32))
                bits[1, 0], bits[1, 1], bits[1, 2] = 0.0, 64.0, 17.0
                h0 = torch.zeros(1, B, h_dim)
                probs = []
                hook = net.enc[5].register_forward_hook(lambda m, i, o: probs.append(o.detach().clone()))
                with torch.no_grad():
                    codes, all_h = net.encode(y, bits, h0)
                    hook.remove()
                    prob = torch.stack(probs).permute(1, 0, 2)
                    mel_hat, h_T = net.decode(codes, h0)
                if margin(prob, bits if var_bit else torch.full((B, T), 64.0)) > nb_thr:
                    break
            print(f"   input seed {in_seed}")
            nb = bits if var_bit else torch.full((B, T), 64.0)
            print(f"   min |p-0.5| over active bits = {margin(prob, nb):.3e}")
            save(f"g3_bvrnn_{tag}", y=y, bits=bits, codes=codes, all_h=all_h, prob=prob,
                 mel_hat=mel_hat, h_T=h_T[0], seed=np.int64(1234))

    # ---------------- G5: BigVGAN ----------------------------------------------------------
    print("G5 BigVGAN")
    gsd = synth.generator_state_dict(conf, seed=1235)
    voc = BigVGAN(AttrDict(conf["vocoder_config"]))
    voc.load_state_dict(gsd)
    voc.eval()
    rng = np.random.default_rng(78)
    mel_in = torch.from_numpy((-4.0 + 1.6 * rng.standard_normal((2, 80, 32))).astype(np.float32))
    outs = {}
    with torch.no_grad():
        for length in (8192, 8192 + 200, 10 ** 9):
            outs[f"wav_{length}"] = voc(mel_in, length)
    print("   wav rms", float(outs["wav_1000000000"].pow(2).mean().sqrt()),
          "max", float(outs["wav_1000000000"].abs().max()))
    save("g5_bigvgan", mel=mel_in, seed=np.int64(1235), **outs)
    # per-stage taps on a smaller case for bisecting
    taps = {}
    hooks = [voc.conv_pre.register_forward_hook(lambda m, i, o: taps.__setitem__("conv_pre", o.detach().clone()))]
    for i in range(4):
        hooks.append(voc.ups[i][1].register_forward_hook(
            lambda m, inp, o, i=i: taps.__setitem__(f"up{i}", o.detach().clone())))
        for j in range(3):
            hooks.append(voc.resblocks[3 * i + j].register_forward_hook(
                lambda m, inp, o, i=i, j=j: taps.__setitem__(f"res{i}_{j}", o.detach().clone())))
    mel_small = mel_in[:1, :, :10].contiguous()
    with torch.no_grad():
        wav_small = voc(mel_small, 10 ** 9)
    for h in hooks:
        h.remove()
    save("g5_bigvgan_taps", mel=mel_small, wav=wav_small, seed=np.int64(1235), **taps)

    # ---------------- G6: end to end through the reference facade ------------------------
    print("G6 end-to-end facade")
    for cfg_path, c, tag in ((ref_cfg_var, conf, "var"), (ref_cfg_64, conf64, "fix")):
        p1, p2 = synth.write_checkpoints(c, tmp, seed=1234, prefix=tag)
        model = BVRNNCodecModel(cfg_path, p1, p2)
        model.eval()
        rates = (3000, 1500, 6000) if tag == "var" else (3000,)
        Lx = 11025 + 77
        for in_seed in range(21, 400, 2):
            x = synth.synthetic_speech(2, Lx, seed=in_seed, kind="noise")
            x[1] = synth.synthetic_speech(1, Lx, seed=in_seed + 1, kind="speech")[0]
            res = {"x": x, "seed": np.int64(1234)}
            worst = 1.0
            with torch.no_grad():
                for br in rates:
                    probs = []
                    hook = model.bvrnn.enc[5].register_forward_hook(
                        lambda m, i, o: probs.append(o.detach().clone()))
                    codes = model.encode(x, br)
                    hook.remove()
                    prob = torch.stack(probs).permute(1, 0, 2)
                    nbits = min(64.0, float(np.round(br * 256 / 22050))) if tag == "var" else 64.0
                    worst = min(worst, margin(prob, torch.full(prob.shape[:2], nbits)))
                    res[f"codes_{br}"] = codes
                    res[f"prob_{br}"] = prob
            if worst > 5e-6:
                break
        print(f"   input seed {in_seed}, min |p-0.5| over active bits = {worst:.3e}")
        with torch.no_grad():
            for br in rates:
                res[f"wav_{br}"] = model.decode(res[f"codes_{br}"], x.shape[1])
            full = model(x, 3000)
            assert torch.equal(full, res["wav_3000"])
            res["wav_untrimmed_3000"] = model.decode(res["codes_3000"], 10 ** 9)
        print(f"   {tag}: wav rms {float(res['wav_3000'].pow(2).mean().sqrt()):.4f}")
        save(f"g6_e2e_{tag}", **res)
    print("done; temporary checkpoints in", tmp)


if __name__ == "__main__":
    main()
