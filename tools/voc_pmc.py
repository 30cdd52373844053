#!/usr/bin/env python3
"""One vocoder pass (B=32, T=215: a quarter of the bench shape, same per-workgroup work) for rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from gpu_common import make_model
model = make_model()[0]
rng = np.random.default_rng(0)
mel = torch.from_numpy((-4 + 1.6 * rng.standard_normal((32, 80, 215))).astype(np.float32)).to("cuda:0")
for i in range(2):
    w = model.vocoder(mel, 55000)
torch.cuda.synchronize()
print("ok", tuple(w.shape))
