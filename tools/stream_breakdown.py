#!/usr/bin/env python3
"""Where a 256-stream 20 ms hop spends its time (host-side wall clock with a sync after each piece)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from gpu_common import make_model
from bvcodec import synth
from bvcodec.streaming import StreamingDecoder, StreamingEncoder
from bvcodec.model import SCALING
B, hop, hops = 256, 441, 200
model = make_model()[0]
x = synth.synthetic_speech(B, hop * hops, seed=3, kind="noise").to("cuda:0")
enc, dec = StreamingEncoder(model, B, 3000), StreamingDecoder(model, B)
acc = {}
def tick(name, t0):
    torch.cuda.synchronize(); t = time.perf_counter(); acc.setdefault(name, []).append(t - t0); return t
for i in range(hops):
    torch.cuda.synchronize(); t = time.perf_counter()
    c = enc.push(x[:, i * hop:(i + 1) * hop]); t = tick("enc.push", t)
    if c.shape[1]:
        mel, dec.h = model.bvrnn.decode(c, dec.h); t = tick("bvrnn.decode", t)
        w = dec.voc.push(mel, SCALING); t = tick("voc.push", t)
# finer: inside enc.push
enc2 = StreamingEncoder(model, B, 3000)
for i in range(hops):
    xx = x[:, i * hop:(i + 1) * hop]
    torch.cuda.synchronize(); t = time.perf_counter()
    enc2.buf = torch.cat([enc2.buf, xx], 1); enc2.n += xx.shape[1]; t = tick("  cat", t)
    if enc2.buf.shape[1] > 2 * enc2.hop:
        mel = model.mel_spectrogram(enc2.buf); t = tick("  mel_spectrogram", t)
for k, v in acc.items():
    v = np.array(v[20:]) * 1e3
    print(f"{k:18s} mean {v.mean():.3f} ms  p50 {np.median(v):.3f}  n={len(v)}")
