"""Helpers shared by the GPU parity tests."""
import os
import tempfile

import numpy as np
import torch

from bvcodec import BVRNNCodecModel, config, synth

_CACHE = {}


def make_model(var_bit=True, h_dim=1024, seed=1234, env=None, mel_stats=None):
    """Product model on cuda:0 with seeded synthetic checkpoints (+ the matching oracle state dicts).
    env: extra environment variables that are read when the engine is created (BVC_NO_GRAPH, ...)."""
    key = (var_bit, h_dim, seed, tuple(sorted((env or {}).items())), mel_stats)
    if key in _CACHE:
        return _CACHE[key]
    base = config.DEFAULT_CONFIG if var_bit else config.DEFAULT_CONFIG_64BIT
    conf = config.load_config(base)
    d = tempfile.mkdtemp(prefix="bvc_test_")
    cfg_path = base
    if h_dim != conf["h_dim"]:
        conf["h_dim"] = h_dim
        cfg_path = os.path.join(d, "cfg.toml")
        with open(base) as f:
            txt = f.read().replace("h_dim = 1024", f"h_dim = {h_dim}")
        with open(cfg_path, "w") as f:
            f.write(txt)
    p1, p2 = synth.write_checkpoints(conf, d, seed=seed, mel_stats=mel_stats)
    model = BVRNNCodecModel(cfg_path, p1, p2).to("cuda:0")
    if env:                      # the library reads its switches in bvc_model_create: create the engine now
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            model.engine()
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    vr = synth.bvrnn_state_dict(conf, seed, mel_stats)
    ge = synth.generator_state_dict(conf, seed + 1)
    _CACHE[key] = (model, conf, vr, ge)
    return _CACHE[key]


def report(name, got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    err = np.abs(got - ref)
    s = (f"{name}: shape {got.shape} max|err| {err.max():.3e} rms err {np.sqrt((err ** 2).mean()):.3e} "
         f"ref rms {np.sqrt((ref ** 2).mean()):.3e} nan {int(np.isnan(got).sum())}")
    print(s, flush=True)
    return err
