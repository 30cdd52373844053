"""Seeded synthetic checkpoints in the reference's checkpoint format.

The shipped checkpoints of the reference are Git-LFS pointers (chkpts/*, .gitattributes:1), so
tests and the benchmark run on synthetic weights written in exactly the layout the facade loads:
``{'vrnn': state_dict}`` and ``{'generator': state_dict}`` (bvrnn_codec_model.py:38-42), with the
state-dict key names of ``BVRNN`` (bvrnn.py:30-83) and ``BigVGAN`` (models.py:132-205, old-style
weight_norm ``weight_g``/``weight_v``).  Values come from numpy's PCG64 stream, so the same seed
gives bit-identical tensors in the build container and on the GPU box (nothing large is shipped).

BVRNN values follow PyTorch's default init ranges (uniform +-1/sqrt(fan_in)), which puts the
encoder logits close to 0 - the worst case for code-bit parity.  Vocoder values are scaled so the
activations stay O(1) through all four stages (the reference's std=0.01 conv init would make the
waveform ~1e-3 and a 1e-4 RMS parity bar meaningless); weight_g is deliberately NOT ||v||, and
alpha/beta, mean_mel/std_mel are non-trivial, so that the folding code is exercised.
"""
import collections
import os

import numpy as np
import torch


def _u(rng, shape, bound):
    return torch.from_numpy(rng.uniform(-bound, bound, size=shape).astype(np.float32))


def _n(rng, shape, std=1.0, mean=0.0):
    return torch.from_numpy((mean + std * rng.standard_normal(size=shape)).astype(np.float32))


def bvrnn_state_dict(conf, seed=1234, mel_stats=None):
    """mel_stats = (mean, std_lo, std_hi): replaces the default conditioning (mean ~ N(-4, 1), std in [0.6, 2.2]) - e.g. (-8.0, 0.05, 0.3),
    the narrow bands a trained checkpoint may carry; every other tensor is the one the same seed gives without it."""
    rng = np.random.default_rng(seed)
    x, h, z = conf["num_mels"], conf["h_dim"], conf["z_dim"]
    sd = collections.OrderedDict()
    sd["mean_mel"] = _n(rng, (x,), 1.0, -4.0)
    sd["std_mel"] = torch.from_numpy(rng.uniform(0.6, 2.2, size=(x,)).astype(np.float32))
    sd["log_sigma"] = torch.tensor([float(conf.get("log_sigma_init", -1.0))], dtype=torch.float32)

    def lin(name, i, o):
        b = 1.0 / np.sqrt(i)
        sd[f"{name}.weight"] = _u(rng, (o, i), b)
        sd[f"{name}.bias"] = _u(rng, (o,), b)

    lin("phi_x.0", x, h); lin("phi_x.2", h, h); lin("phi_x.4", h, h)
    lin("phi_z.0", z, h); lin("phi_z.2", h, h); lin("phi_z.4", h, h)
    lin("enc.0", 2 * h, h); lin("enc.2", h, h); lin("enc.4", h, z)
    lin("prior.0", h, h); lin("prior.2", h, h); lin("prior.4", h, z)
    lin("dec.0", 2 * h, h); lin("dec.2", h, h); lin("dec.4", h, h); lin("dec.6", h, x)
    b = 1.0 / np.sqrt(h)
    sd["rnn.weight_ih_l0"] = _u(rng, (3 * h, 2 * h), b)
    sd["rnn.weight_hh_l0"] = _u(rng, (3 * h, h), b)
    sd["rnn.bias_ih_l0"] = _u(rng, (3 * h,), b)
    sd["rnn.bias_hh_l0"] = _u(rng, (3 * h,), b)
    if mel_stats is not None:
        r2 = np.random.default_rng(seed + 7919)
        mean, lo, hi = mel_stats
        sd["mean_mel"] = _n(r2, (x,), 0.5, mean)
        sd["std_mel"] = torch.from_numpy(r2.uniform(lo, hi, size=(x,)).astype(np.float32))
    return sd


def generator_state_dict(conf, seed=4321):
    rng = np.random.default_rng(seed)
    v = conf["vocoder_config"]
    sd = collections.OrderedDict()

    def conv(name, cout, cin, k, gain):
        sd[f"{name}.bias"] = _u(rng, (cout,), 0.1)
        sd[f"{name}.weight_g"] = torch.from_numpy(
            (gain * rng.uniform(0.8, 1.2, size=(cout, 1, 1))).astype(np.float32))
        sd[f"{name}.weight_v"] = _n(rng, (cout, cin, k))

    def convt(name, cin, cout, k, stride, gain):
        sd[f"{name}.bias"] = _u(rng, (cout,), 0.1)
        g = gain * np.sqrt(cout * k / (cin * k / stride))
        sd[f"{name}.weight_g"] = torch.from_numpy(
            (g * rng.uniform(0.8, 1.2, size=(cin, 1, 1))).astype(np.float32))
        sd[f"{name}.weight_v"] = _n(rng, (cin, cout, k))

    c0 = v["upsample_initial_channel"]
    conv("conv_pre", c0, v["num_mels"], 7, 0.25)          # mel values are O(4): keep y0 O(1)
    nk = len(v["resblock_kernel_sizes"])
    ch = c0
    for i, (u, k) in enumerate(zip(v["upsample_rates"], v["upsample_kernel_sizes"])):
        convt(f"ups.{i}.1", ch, ch // 2, k, u, 0.7)
        ch //= 2
    ch = c0
    for i in range(len(v["upsample_rates"])):
        ch //= 2
        for j, ks in enumerate(v["resblock_kernel_sizes"]):
            pre = f"resblocks.{i * nk + j}"
            for m in range(3):
                conv(f"{pre}.convs1.{m}", ch, ch, ks, 0.6)
            for m in range(3):
                conv(f"{pre}.convs2.{m}", ch, ch, ks, 0.35)
            for a in range(6):
                sd[f"{pre}.activations.{a}.alpha"] = _n(rng, (ch,), 0.3)
                sd[f"{pre}.activations.{a}.beta"] = _n(rng, (ch,), 0.3)
    sd["activation_post.alpha"] = _n(rng, (ch,), 0.3)
    sd["activation_post.beta"] = _n(rng, (ch,), 0.3)
    conv("conv_post", 1, ch, 7, 0.25)
    return sd


def write_checkpoints(conf, directory, seed=1234, prefix="synthetic", mel_stats=None):
    """Write both checkpoints in the reference format; returns (bvrnn_path, vocoder_path)."""
    os.makedirs(directory, exist_ok=True)
    p1 = os.path.join(directory, f"{prefix}_bvrnn_h{conf['h_dim']}_seed{seed}{'_melstats' if mel_stats else ''}")
    p2 = os.path.join(directory, f"{prefix}_bigvgan_seed{seed}")
    torch.save({"vrnn": bvrnn_state_dict(conf, seed, mel_stats)}, p1)
    torch.save({"generator": generator_state_dict(conf, seed + 1)}, p2)
    return p1, p2


def synthetic_speech(batch, length, seed=0, kind="noise", fs=22050):
    """Synthetic inputs of SURVEY.md 8(d): 'noise' = 0.1*N(0,1) clipped to [-1,1];
    'speech' = 5 harmonics with 4 Hz AM plus noise, peak-normalised (example.py:17)."""
    rng = np.random.default_rng(seed)
    if kind == "noise":
        x = np.clip(0.1 * rng.standard_normal(size=(batch, length)), -1.0, 1.0)
    else:
        t = np.arange(length) / fs
        x = np.zeros((batch, length))
        for b in range(batch):
            f0 = rng.uniform(90.0, 250.0)
            am = 0.55 + 0.45 * np.sin(2 * np.pi * 4.0 * t + rng.uniform(0, 6.28))
            for hm in range(1, 6):
                x[b] += (1.0 / hm) * np.sin(2 * np.pi * f0 * hm * t + rng.uniform(0, 6.28))
            x[b] = x[b] * am + 0.02 * rng.standard_normal(length)
            x[b] /= np.max(np.abs(x[b]))
    return torch.from_numpy(x.astype(np.float32))
