#!/usr/bin/env python3
"""Vocoder only at the BASELINE configs[1] shape (B=64, T=430) for rocprofv3 --kernel-trace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from gpu_common import make_model
model = make_model()[0]
rng = np.random.default_rng(0)
mel = torch.from_numpy((-4 + 1.6 * rng.standard_normal((64, 80, 430))).astype(np.float32)).to("cuda:0")
for i in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    w = model.vocoder(mel, 110250)
    torch.cuda.synchronize(); print(f"vocoder {1e3*(time.time()-t0):.2f} ms", flush=True)
