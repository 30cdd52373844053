#!/usr/bin/env python3
"""BASELINE.json configs[4]: 256 concurrent streams, 20 ms (441-sample) hops, config_varBitRate @ 3 kbit/s,
per-hop encode+decode on one MI355X.  Prints one JSON line with p50 / p99 per-hop latency."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from gpu_common import make_model
from bvcodec import synth
from bvcodec.streaming import StreamingCodec, StreamingDecoder, StreamingEncoder

B, hop, hops = 256, 441, 300
incremental = "--context" not in sys.argv      # --context: stateless vocoder that re-runs a 26-frame context per hop
python_path = "--python" in sys.argv or not incremental     # --python: the round-1 per-hop schedule driven from Python
model = make_model()[0]
x = synth.synthetic_speech(B, hop * hops, seed=3, kind="noise").to("cuda:0")
lat, frames = [], 0
if python_path:
    enc, dec = StreamingEncoder(model, B, 3000), StreamingDecoder(model, B, incremental=incremental)
    for i in range(hops):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c = enc.push(x[:, i * hop:(i + 1) * hop])
        w = dec.push(c)
        torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
        frames += c.shape[1]
else:                                          # default: one library call per hop, replayed from a hipGraph once warm
    sc = StreamingCodec(model, B, 3000, hop=hop)
    for i in range(hops):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c, w = sc.push(x[:, i * hop:(i + 1) * hop])
        torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
        frames += c.shape[1]
    model.check_status()
lat = np.array(lat[50:]) * 1e3
print(json.dumps({"config": "BASELINE configs[4]: 256 streams x 20 ms hops @ 3 kbit/s, per-hop encode+decode",
                  "schedule": "python-driven hop (round 1)" if python_path else ("bvc_stream_codec_tick: whole hop in one call, persistent recurrence" if os.environ.get("BVC_STREAM_FLOW") != "0" else ("bvc_stream_codec_tick: launch-per-layer recurrence, hipGraph-replayed" if os.environ.get("BVC_STREAM_NO_GRAPH") != "1" else "bvc_stream_codec_tick: launch-per-layer recurrence, eager launches")),
                  "vocoder": "incremental (history buffers)" if incremental else "context recompute (26 frames)",
                  "p50_ms": round(float(np.percentile(lat, 50)), 3), "p99_ms": round(float(np.percentile(lat, 99)), 3),
                  "mean_ms": round(float(lat.mean()), 3), "hop_budget_ms": 20.0, "frames_per_hop": round(frames / hops, 3),
                  "real_time_factor_per_stream": round(20.0 / float(lat.mean()), 2)}))
