#!/usr/bin/env python3
"""Checks bvcodec.dist.concurrent_streams against the decode-chain concurrency matrix (tools/queue_map_probe.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bvcodec import dist as bdist
if os.environ.get("BVC_FORCE_PG") == "1":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
    bdist.init_from_env()
    t = torch.ones(4, device="cuda:0"); torch.distributed.all_reduce(t)
dev = torch.device("cuda:0")
t0 = time.perf_counter()
ss = bdist.concurrent_streams(3, dev)
print("picked", [hex(s.cuda_stream) for s in ss], f"in {time.perf_counter()-t0:.3f} s")
