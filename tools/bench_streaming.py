#!/usr/bin/env python3
"""BASELINE.json configs[4]: 256 concurrent streams, 20 ms (441-sample) hops, config_varBitRate @ 3 kbit/s,
per-hop encode+decode on one MI355X.  Prints one JSON line with p50 / p99 per-hop latency."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from gpu_common import make_model
from bvcodec import synth
from bvcodec.streaming import StreamingDecoder, StreamingEncoder

B, hop, hops = 256, 441, 300
incremental = "--context" not in sys.argv      # --context: stateless vocoder that re-runs a 26-frame context per hop
model = make_model()[0]
x = synth.synthetic_speech(B, hop * hops, seed=3, kind="noise").to("cuda:0")
enc, dec = StreamingEncoder(model, B, 3000), StreamingDecoder(model, B, incremental=incremental)
lat, frames = [], 0
for i in range(hops):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c = enc.push(x[:, i * hop:(i + 1) * hop])
    w = dec.push(c)
    torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
    frames += c.shape[1]
lat = np.array(lat[50:]) * 1e3
print(json.dumps({"config": "BASELINE configs[4]: 256 streams x 20 ms hops @ 3 kbit/s, per-hop encode+decode",
                  "vocoder": "incremental (history buffers)" if incremental else "context recompute (26 frames)",
                  "p50_ms": round(float(np.percentile(lat, 50)), 3), "p99_ms": round(float(np.percentile(lat, 99)), 3),
                  "mean_ms": round(float(lat.mean()), 3), "hop_budget_ms": 20.0, "frames_per_hop": round(frames / hops, 3),
                  "real_time_factor_per_stream": round(20.0 / float(lat.mean()), 2)}))
