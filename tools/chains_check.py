#!/usr/bin/env python3
"""Batches of more than 64 utterances on the persistent recurrence (interleaved chains, k_flow.hip MULTI) against the
launch-per-layer schedule: same codes, same waveform; and the time per step of both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from gpu_common import make_model
from bvcodec import synth

model = make_model()[0]
for B, secs in ((80, 0.5), (128, 0.5), (256, 0.5), (250, 1.0)):
    L = int(22050 * secs)
    x = synth.synthetic_speech(B, L, seed=B, kind="speech").to("cuda:0")
    model.set_recurrence("persistent")
    c1 = model.encode(x, 3000); w1 = model.decode(c1, L); torch.cuda.synchronize(); model.check_status()
    model.set_recurrence("layers")
    c2 = model.encode(x, 3000); w2 = model.decode(c1, L); torch.cuda.synchronize()
    print(f"B={B} T={c1.shape[1]}: code bits differing {int((c1 != c2).sum())} of {c1.numel()}, wav max diff {float((w1 - w2).abs().max()):.3e}", flush=True)
if "--time" in sys.argv:
    for B in (128, 256):
        L = 22050 * 5
        x = synth.synthetic_speech(B, L, seed=1, kind="noise").to("cuda:0")
        for mode in ("persistent", "layers"):
            model.set_recurrence(mode)
            for _ in range(2):
                c = model.encode(x, 3000); w = model.decode(c, L)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                c = model.encode(x, 3000); w = model.decode(c, L)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            model.check_status()
            print(f"B={B} x 5 s {mode}: {dt * 1e3:.1f} ms/step, {B * 5 / dt:.0f} audio-s/s", flush=True)
