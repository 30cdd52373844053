"""Checkpoint loading: reference checkpoint dicts -> named float32 host tensors for the C ABI.

Accepts exactly what the reference facade loads (bvrnn_codec_model.py:38-42):
``torch.load(path, weights_only=True)`` dicts ``{'vrnn': state_dict}`` / ``{'generator': state_dict}``
with the key names of ``BVRNN`` (bvrnn.py:30-83) and ``BigVGAN`` (models.py:132-205).  Loading is
strict like ``load_state_dict``: missing or unexpected keys raise ``RuntimeError``.

The old-style weight norm of the generator (``weight_g``/``weight_v``, models.py:47-62,140,164,200)
is folded ONCE here with the same ATen function the reference's forward pre-hook calls on every
forward (``torch._weight_norm(v, g, 0)``), so the folded weights are bit-identical to the ones the
reference convolves with.  The re-layout into MFMA fragment order happens inside the library.
"""
import torch

from . import melbank


def expected_bvrnn_keys(conf):
    keys = ["mean_mel", "std_mel", "log_sigma"]
    for net, idx in (("phi_x", (0, 2, 4)), ("phi_z", (0, 2, 4)), ("enc", (0, 2, 4)),
                     ("prior", (0, 2, 4)), ("dec", (0, 2, 4, 6))):
        for i in idx:
            keys += [f"{net}.{i}.weight", f"{net}.{i}.bias"]
    keys += ["rnn.weight_ih_l0", "rnn.weight_hh_l0", "rnn.bias_ih_l0", "rnn.bias_hh_l0"]
    return keys


def expected_generator_keys(conf):
    v = conf["vocoder_config"]
    keys = []

    def wn(name):
        keys.extend([f"{name}.bias", f"{name}.weight_g", f"{name}.weight_v"])

    wn("conv_pre")
    nk = len(v["resblock_kernel_sizes"])
    for i in range(len(v["upsample_rates"])):
        wn(f"ups.{i}.1")
    for n in range(len(v["upsample_rates"]) * nk):
        for m in range(3):
            wn(f"resblocks.{n}.convs1.{m}")
        for m in range(3):
            wn(f"resblocks.{n}.convs2.{m}")
        for a in range(6):
            keys += [f"resblocks.{n}.activations.{a}.alpha", f"resblocks.{n}.activations.{a}.beta"]
    keys += ["activation_post.alpha", "activation_post.beta"]
    wn("conv_post")
    return keys


def _strict(sd, expected, what):
    missing = [k for k in expected if k not in sd]
    unexpected = [k for k in sd if k not in set(expected)]
    if missing or unexpected:
        raise RuntimeError(f"Error(s) in loading state_dict for {what}: "
                           f"Missing key(s): {missing[:8]}{'...' if len(missing) > 8 else ''}; "
                           f"Unexpected key(s): {unexpected[:8]}{'...' if len(unexpected) > 8 else ''}")


def load_checkpoint(path, top_key):
    chk = torch.load(path, map_location=torch.device("cpu"), weights_only=True)
    if top_key not in chk:
        raise KeyError(top_key)
    return chk[top_key]


def host_tensors(conf, vrnn_sd, gen_sd):
    """-> dict name -> contiguous float32 CPU tensor, the set bvc_model_create expects."""
    _strict(vrnn_sd, expected_bvrnn_keys(conf), "BVRNN")
    _strict(gen_sd, expected_generator_keys(conf), "BigVGAN")
    out = {}
    for k, t in vrnn_sd.items():
        if k == "log_sigma":                                # only used by the training loss (bvrnn.py:33)
            continue
        out[k] = t
    for k, t in gen_sd.items():
        if k.endswith(".weight_g"):
            base = k[: -len(".weight_g")]
            out[base + ".weight"] = torch._weight_norm(gen_sd[base + ".weight_v"].float(), t.float(), 0)
        elif k.endswith(".weight_v"):
            continue
        else:
            out[k] = t
    out["mel_basis"] = torch.from_numpy(melbank.slaney_mel_basis(conf["fs"], conf["winsize"], conf["num_mels"],
                                                                conf["fmin"], conf["fmax"]))
    out["hann_window"] = torch.hann_window(conf["winsize"], dtype=torch.float32)     # meldataset.py:70
    return {k: t.detach().to(torch.float32).contiguous() for k, t in out.items()}
