#!/usr/bin/env python3
"""Generate tests/golden/g8_bvrnn_forward_*.npz by running the REFERENCE's ``BVRNN.forward`` (bvrnn.py:86-160)
in the build container.  Only torch is needed by that module.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_forward.py [--ref /root/reference]

The reference draws its randomness from the global CPU generator (one scalar per frame, plus one (B, z_dim)
uniform tensor per frame unless greedy).  The script seeds the generator, runs the reference, re-seeds and
re-draws the same sequence (oracle.bvrnn.draw_randomness) so that the fixture can store the numbers; it
asserts that the oracle fed with those numbers reproduces the reference output, which proves the stored
sequence is the one the reference consumed.  Weights are regenerated from the seed by the tests.
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from bvcodec import config as bconfig, synth          # noqa: E402
from oracle import bvrnn as obv                       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    a = ap.parse_args()
    sys.dont_write_bytecode = True
    sys.path.insert(0, a.ref)
    torch.set_num_threads(8)
    from bvrnn import BVRNN                                              # reference

    conf = bconfig.load_config(os.path.join(a.ref, "configs", "config_varBitRate.toml"))
    B, T = 2, 24
    for h_dim, var_bit in ((1024, True), (1024, False), (64, True)):
        c = dict(conf); c["h_dim"] = h_dim; c["var_bit"] = var_bit
        tag = f"h{h_dim}_{'var' if var_bit else 'fix'}"
        sd = synth.bvrnn_state_dict(c, seed=1234)
        net = BVRNN(80, h_dim, 64, [np.zeros(80), np.ones(80)], c["log_sigma_init"], variableBit=var_bit)
        net.load_state_dict(sd)
        net.eval()
        net.device = torch.device("cpu")
        out = {"seed": np.int64(1234)}
        for mode, (p_use_gen, greedy) in enumerate(((0.0, True), (1.0, False), (0.5, False), (0.5, True))):
            for rs in range(100 + 1000 * mode, 100 + 1000 * mode + 600):
                rng = np.random.default_rng(rs)
                y = torch.from_numpy((-4.0 + 1.6 * rng.standard_normal((B, T, 80))).astype(np.float32))
                bits = torch.full((B, T), 35.0)
                bits[1] = torch.from_numpy(rng.integers(0, 65, size=T).astype(np.float32))
                bits[1, 0], bits[1, 1] = 0.0, 64.0
                probs, priors = [], []
                hk = [net.enc[5].register_forward_hook(lambda m, i, o: probs.append(o.detach().clone())),
                      net.prior[5].register_forward_hook(lambda m, i, o: priors.append(o.detach().clone()))]
                torch.manual_seed(rs)
                with torch.no_grad():
                    dec, kld = net(y, p_use_gen, greedy, bits)
                for h in hk:
                    h.remove()
                torch.manual_seed(rs)
                r, noise = obv.draw_randomness(T, B, 64, greedy)
                o = obv.forward(sd, y, p_use_gen, greedy, bits, r, noise, var_bit=var_bit)
                prob = torch.stack(probs).permute(1, 0, 2)
                prior = torch.stack(priors).permute(1, 0, 2)
                # the stored random numbers are the ones the reference consumed: same states selected, same samples
                assert torch.allclose(o["prob"], prob, atol=2e-6), (tag, mode, float((o["prob"] - prob).abs().max()))
                assert torch.allclose(o["dec"], dec, atol=2e-5), (tag, mode, float((o["dec"] - dec).abs().max()))
                # tie margin of the rounding (active bits only), evaluated in float64
                o64 = obv.forward(sd, y, p_use_gen, greedy, bits, r, noise, var_bit=var_bit, dtype=torch.float64)
                act = (torch.arange(64)[None, None, :] < (bits if var_bit else torch.full((B, T), 64.0))[:, :, None])
                marg = float((o64["arg"] - 0.5).abs()[act].min())
                same = bool(torch.equal(torch.round(o64["arg"]).float()[act], torch.round(o["arg"])[act]))
                if marg > 1.5e-5 and same:
                    break
            print(f"{tag} mode {mode} (p_use_gen={p_use_gen}, greedy={greedy}): seed {rs}, rounding margin {marg:.2e}, "
                  f"kld {float(kld):.6f}, use_gen {int((r < p_use_gen).sum())}/{T}")
            k = f"m{mode}_"
            out.update({k + "y": y, k + "bits": bits, k + "p_use_gen": np.float64(p_use_gen), k + "greedy": np.bool_(greedy),
                        k + "r": r, k + "dec": dec, k + "kld": kld, k + "prob": prob, k + "prior": prior,
                        k + "torch_seed": np.int64(rs)})
            if noise is not None:
                out[k + "noise"] = noise
        path = os.path.join(HERE, f"g8_bvrnn_forward_{tag}.npz")
        np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                     for k, v in out.items()})
        print(f"  wrote {os.path.basename(path)} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
