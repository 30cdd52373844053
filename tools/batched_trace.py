#!/usr/bin/env python3
"""One encode + decode at the bench shape for rocprofv3 --kernel-trace: per-dispatch durations of the batched GEMMs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from gpu_common import make_model
from bvcodec import synth
model = make_model()[0]
x = synth.synthetic_speech(64, 110250, seed=0, kind="noise").to("cuda:0")
for _ in range(2):
    codes = model.encode(x, 3000)
    mel, _ = model.bvrnn.decode(codes, torch.zeros(1, 64, 1024, device="cuda:0"))
torch.cuda.synchronize()
print("ok")
