"""Oracle: BVRNN mel coder, greedy encode / decode (test infrastructure, see __init__).

Follows ``BVRNN.encode`` bvrnn.py:163-209 and ``BVRNN.decode`` bvrnn.py:211-229 with the
networks declared at bvrnn.py:44-83.  Weights come in as the reference ``state_dict``
(keys ``phi_x.0.weight`` ... ``rnn.bias_hh_l0``).  The single-step ``nn.GRU`` call
(bvrnn.py:206,227) is written out as the PyTorch GRU cell:
    gi = W_ih x + b_ih ; gh = W_hh h + b_hh ; r = sigmoid(gh_r + gi_r) ; z = sigmoid(gh_z + gi_z)
    n = tanh(gi_n + r * gh_n) ; h' = (h - n) * z + n
"""
import torch
import torch.nn.functional as F


def _mlp(sd, name, idx, x, last_act=True):
    for j, i in enumerate(idx):
        x = F.linear(x, sd[f"{name}.{i}.weight"], sd[f"{name}.{i}.bias"])
        if last_act or j + 1 < len(idx):
            x = F.elu(x)
    return x


def phi_x(sd, y):          # bvrnn.py:44-50
    return _mlp(sd, "phi_x", (0, 2, 4), y)


def phi_z(sd, z):          # bvrnn.py:52-58
    return _mlp(sd, "phi_z", (0, 2, 4), z)


def enc_logits(sd, u):     # bvrnn.py:60-65 without the final Sigmoid
    return _mlp(sd, "enc", (0, 2, 4), u, last_act=False)


def dec(sd, u):            # bvrnn.py:77-80
    return _mlp(sd, "dec", (0, 2, 4, 6), u, last_act=False)


def gru_cell(sd, x, h):    # nn.GRU(2h, h, 1) stepped once; bvrnn.py:83
    gi = F.linear(x, sd["rnn.weight_ih_l0"], sd["rnn.bias_ih_l0"])
    gh = F.linear(h, sd["rnn.weight_hh_l0"], sd["rnn.bias_hh_l0"])
    i_r, i_z, i_n = gi.chunk(3, 1)
    h_r, h_z, h_n = gh.chunk(3, 1)
    r = torch.sigmoid(h_r + i_r)
    z = torch.sigmoid(h_z + i_z)
    n = torch.tanh(i_n + r * h_n)
    return (h - n) * z + n


def cast_state(sd, dtype):
    return {k: v.to(dtype) for k, v in sd.items()}


@torch.no_grad()
def encode(sd, y, bits_per_frame, h0, var_bit=True, dtype=torch.float32, forced_h=None):
    """y (B,T,80) mel, bits_per_frame (B,T), h0 (B,H).

    Returns dict(codes (B,T,Z), all_h (B,T,H) [state BEFORE frame t], prob (B,T,Z)
    [sigmoid output before rounding], logit (B,T,Z), h_last (B,H)).
    ``forced_h`` (B,T,H): teacher-forced mode, frame t starts from forced_h[:, t].
    """
    sd = cast_state(sd, dtype)
    y = torch.as_tensor(y).to(dtype)
    h = torch.as_tensor(h0).to(dtype)
    bits = torch.as_tensor(bits_per_frame).to(dtype)
    mean, std = sd["mean_mel"], sd["std_mel"]
    yn = (y - mean[None, None, :]) / std[None, None, :]              # bvrnn.py:173
    px = phi_x(sd, yn)                                               # bvrnn.py:178
    zdim = sd["enc.4.weight"].shape[0]
    if var_bit:                                                      # bvrnn.py:180-182
        mask = (bits[:, :, None] > torch.arange(zdim, dtype=dtype)[None, None, :]).to(dtype)
    codes, all_h, probs, logits = [], [], [], []
    for t in range(y.shape[1]):                                      # bvrnn.py:186
        if forced_h is not None:
            h = torch.as_tensor(forced_h[:, t]).to(dtype)
        logit = enc_logits(sd, torch.cat([px[:, t], h], 1))          # bvrnn.py:189
        p = torch.sigmoid(logit)
        z = torch.round(p)                                           # bvrnn.py:191
        if var_bit:                                                  # bvrnn.py:193-194
            z = z * mask[:, t] + 0.5 * (1 - mask[:, t])
        codes.append(z); probs.append(p); logits.append(logit); all_h.append(h)
        pz = phi_z(sd, z)                                            # bvrnn.py:198
        d = dec(sd, torch.cat([pz, h], 1))                           # bvrnn.py:202
        pxg = phi_x(sd, (d - mean[None, :]) / std[None, :])          # bvrnn.py:204
        h = gru_cell(sd, torch.cat([pxg, pz], 1), h)                 # bvrnn.py:206
    st = lambda l: torch.stack(l).permute(1, 0, 2).contiguous()
    return dict(codes=st(codes), all_h=st(all_h), prob=st(probs), logit=st(logits), h_last=h)


@torch.no_grad()
def decode(sd, z, h0, dtype=torch.float32):
    """z (B,T,Z) codes, h0 (B,H) -> dict(mel (B,T,80), h_last (B,H))."""
    sd = cast_state(sd, dtype)
    z = torch.as_tensor(z).to(dtype)
    h = torch.as_tensor(h0).to(dtype)
    mean, std = sd["mean_mel"], sd["std_mel"]
    out = []
    for t in range(z.shape[1]):                                      # bvrnn.py:222
        pz = phi_z(sd, z[:, t])
        d = dec(sd, torch.cat([pz, h], 1))
        out.append(d)
        pxg = phi_x(sd, (d - mean[None, :]) / std[None, :])
        h = gru_cell(sd, torch.cat([pxg, pz], 1), h)
    return dict(mel=torch.stack(out).permute(1, 0, 2).contiguous(), h_last=h)


def prior_prob(sd, h):     # bvrnn.py:68-73 (with the final Sigmoid)
    return torch.sigmoid(_mlp(sd, "prior", (0, 2, 4), h, last_act=False))


def draw_randomness(T, B, zdim, greedy, generator=None):
    """The random numbers ``BVRNN.forward`` consumes, in its order (bvrnn.py:111,129): per frame one scalar
    ``torch.rand([])`` and, unless greedy, one ``torch.rand_like(enc_t)`` of shape (B, z_dim).  Drawn from
    the given (default: global CPU) generator.  Returns (r (T,), noise (B,T,Z) or None)."""
    r, noise = [], []
    for _ in range(T):
        r.append(torch.rand([], generator=generator))
        if not greedy:
            noise.append(torch.rand(B, zdim, generator=generator))
    return torch.stack(r), (None if greedy else torch.stack(noise, 1).contiguous())


@torch.no_grad()
def forward(sd, y, p_use_gen, greedy, bits_per_frame, r, noise, var_bit=True, dtype=torch.float32):
    """``BVRNN.forward`` (bvrnn.py:86-160), forward values only.  r (T,) are the per-frame scalars compared
    against p_use_gen, noise (B,T,Z) the uniform samples of the Bernoulli sampler (None when greedy).

    Returns dict(dec (B,T,80), kld scalar, kld_frames (T,), z (B,T,Z) [forward value of the straight-through
    sample], prob (B,T,Z) [enc_t], prior (B,T,Z), arg (B,T,Z) [the value that is rounded])."""
    sd = cast_state(sd, dtype)
    y = torch.as_tensor(y).to(dtype)
    mean, std = sd["mean_mel"], sd["std_mel"]
    yn = (y - mean[None, None, :]) / std[None, None, :]              # bvrnn.py:96
    px = phi_x(sd, yn)                                               # bvrnn.py:101
    B, T = y.shape[:2]
    H = sd["rnn.weight_hh_l0"].shape[1]
    zdim = sd["enc.4.weight"].shape[0]
    h = torch.zeros(B, H, dtype=dtype)                               # bvrnn.py:103-104
    h2 = torch.zeros(B, H, dtype=dtype)
    if var_bit:                                                      # bvrnn.py:105-107
        bits = torch.as_tensor(bits_per_frame).to(dtype)
        mask = (bits[:, :, None] > torch.arange(zdim, dtype=dtype)[None, None, :]).to(dtype)
    outs = dict(dec=[], z=[], prob=[], prior=[], arg=[])
    kld = []
    for t in range(T):                                               # bvrnn.py:110
        hs = h2 if bool(r[t] < p_use_gen) else h                     # bvrnn.py:115-120
        e = torch.sigmoid(enc_logits(sd, torch.cat([px[:, t], hs], 1)))
        pr = prior_prob(sd, hs)
        if greedy:                                                   # bvrnn.py:123-126 (straight-through value)
            arg = e
        else:
            arg = torch.as_tensor(noise[:, t]).to(dtype) - 0.5 + e
        z = torch.round(arg) - e + e
        if var_bit:                                                  # bvrnn.py:128-129
            z = z * mask[:, t] + 0.5 * (1 - mask[:, t])
        pz = phi_z(sd, z)                                            # bvrnn.py:131
        d = dec(sd, torch.cat([pz, hs], 1))                          # bvrnn.py:134-137
        pxg = phi_x(sd, (d - mean[None, :]) / std[None, :])          # bvrnn.py:139
        if p_use_gen < 1:                                            # bvrnn.py:142-145
            h = gru_cell(sd, torch.cat([px[:, t], pz], 1), h)
        if p_use_gen > 0:
            h2 = gru_cell(sd, torch.cat([pxg, pz], 1), h2)
        ke = e * (torch.log(torch.clip(e, 1e-3)) - torch.log(torch.clip(pr, 1e-3))) + \
            (1 - e) * (torch.log(torch.clip(1 - e, 1e-3)) - torch.log(torch.clip(1 - pr, 1e-3)))   # bvrnn.py:148-149
        kld.append(torch.mean(torch.sum(ke * mask[:, t], -1)) if var_bit else torch.mean(torch.sum(ke, -1)))
        for k, v in (("dec", d), ("z", z), ("prob", e), ("prior", pr), ("arg", arg)):
            outs[k].append(v)
    res = {k: torch.stack(v).permute(1, 0, 2).contiguous() for k, v in outs.items()}
    res["kld_frames"] = torch.stack(kld)
    res["kld"] = torch.mean(res["kld_frames"])                       # bvrnn.py:160
    return res
