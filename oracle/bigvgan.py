"""Oracle: causal-tiny BigVGAN generator forward (test infrastructure, see __init__).

Follows ``BigVGAN.forward`` third_party/BigVGAN/models.py:207-238 (ctor :132-205),
``AMPBlock1.forward`` models.py:103-121 (causal paddings :41-44, ``get_padding_causal``
:19-20), ``SnakeBeta.forward`` third_party/BigVGAN/activations.py:107-120 with
``alpha_logscale=True``, and old-style ``weight_norm`` (``weight_g``/``weight_v``,
norm over every dim but 0; models.py:47-62,140,164,200).  Only the configuration the two
shipped TOMLs select is covered: resblock "1", snakebeta, no anti-aliasing, all layers
causal (configs/config_varBitRate.toml:39-56).
"""
import torch
import torch.nn.functional as F


def fold_weight_norm(g, v):
    """w = g * v / ||v||, norm over all dims except 0 (torch._weight_norm, dim=0)."""
    norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape([-1] + [1] * (v.dim() - 1))
    return v * (g / norm)


def _w(sd, name):
    return fold_weight_norm(sd[name + ".weight_g"], sd[name + ".weight_v"])


def snakebeta(x, alpha, beta):
    a = torch.exp(alpha)[None, :, None]                      # activations.py:111-115
    b = torch.exp(beta)[None, :, None]
    return x + (1.0 / (b + 0.000000001)) * torch.pow(torch.sin(x * a), 2)   # :116


def amp_block(sd, pre, x, ksize, dilations=(1, 3, 5)):
    """AMPBlock1.forward, models.py:103-121 (symmetric=False)."""
    for m, d in enumerate(dilations):
        a1 = (sd[f"{pre}.activations.{2 * m}.alpha"], sd[f"{pre}.activations.{2 * m}.beta"])
        a2 = (sd[f"{pre}.activations.{2 * m + 1}.alpha"], sd[f"{pre}.activations.{2 * m + 1}.beta"])
        xt = snakebeta(x, *a1)
        xt = F.pad(xt, (ksize * d - d, 0))
        xt = F.conv1d(xt, _w(sd, f"{pre}.convs1.{m}"), sd[f"{pre}.convs1.{m}.bias"], dilation=d)
        xt = snakebeta(xt, *a2)
        xt = F.pad(xt, (ksize - 1, 0))
        xt = F.conv1d(xt, _w(sd, f"{pre}.convs2.{m}"), sd[f"{pre}.convs2.{m}.bias"])
        x = xt + x
    return x


@torch.no_grad()
def forward(sd, cfg, mel, length, dtype=torch.float32, taps=None):
    """mel (B, num_mels, T) -> (B, 1, min(length, 256T+294)).

    ``cfg``: the ``vocoder_config`` table of the TOML.  ``taps``: optional dict that is
    filled with the intermediate tensors (conv_pre, up{i}, stage{i}) for bisecting.
    """
    sd = {k: v.to(dtype) for k, v in sd.items()}
    x = torch.as_tensor(mel).to(dtype)
    rates, ksz = cfg["upsample_rates"], cfg["upsample_kernel_sizes"]
    rks, rds = cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"]
    nk = len(rks)
    x = F.pad(x, [6, 0])                                             # models.py:212
    x = F.conv1d(x, _w(sd, "conv_pre"), sd["conv_pre.bias"])         # :213
    if taps is not None:
        taps["conv_pre"] = x
    for i in range(len(rates)):                                      # :214
        x = F.conv_transpose1d(x, _w(sd, f"ups.{i}.1"), sd[f"ups.{i}.1.bias"],
                               stride=rates[i], padding=0)           # :216-217
        if taps is not None:
            taps[f"up{i}"] = x
        xs = None
        for j in range(nk):                                          # :219-224
            r = amp_block(sd, f"resblocks.{i * nk + j}", x, rks[j], tuple(rds[j]))
            xs = r if xs is None else xs + r
        x = xs / nk                                                  # :225
        if taps is not None:
            taps[f"stage{i}"] = x
    x = snakebeta(x, sd["activation_post.alpha"], sd["activation_post.beta"])   # :228
    x = F.pad(x, [6, 0])                                             # :233
    x = F.conv1d(x, _w(sd, "conv_post"), sd["conv_post.bias"])       # :235
    x = torch.tanh(x)                                                # :236
    return x[:, :, :length]                                          # :238
