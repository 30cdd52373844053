#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: B = argv[2] (default 64), `frames` frames (argv[1], default 12; 430 = configs[1]) of
encode + decode twice, so that every kernel family appears with its configs[1] per-launch shape.  (BVC_NO_GRAPH only
matters for the launch-per-layer schedule, BVC_RECURRENCE=layers.)"""
import os
import sys

os.environ["BVC_NO_GRAPH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch                                        # noqa: E402
from gpu_common import make_model                   # noqa: E402
from bvcodec import synth                           # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 12
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
model = make_model()[0]
x = synth.synthetic_speech(batch, 256 * frames + 40, seed=0, kind="noise").to("cuda:0")
for _ in range(2):
    codes = model.encode(x, 3000)
    wav = model.decode(codes, x.shape[1])
torch.cuda.synchronize()
print("ok", tuple(codes.shape), tuple(wav.shape))
