#!/usr/bin/env python3
"""Which torch streams run concurrently?  For each pair of candidate streams, time two identical dependent chains
(BVRNN decode, T frames) issued on the pair against one chain alone.  ratio ~1: concurrent; ~2: serialised (same
hardware queue).  Env: GPU_MAX_HW_QUEUES, BVC_FORCE_PG=1 (initialises a 1-rank RCCL process group first)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from gpu_common import make_model
from bvcodec import dist as bdist
if os.environ.get("BVC_FORCE_PG") == "1":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
    bdist.init_from_env()
    t = torch.ones(4, device="cuda:0"); torch.distributed.all_reduce(t)
model = make_model()[0]
dev = torch.device("cuda:0")
N = int(os.environ.get("NCAND", "8"))
cands = [torch.cuda.Stream(dev) for _ in range(N)]
codes = (torch.rand(64, 24, 64, device=dev) > 0.5).float()
h0 = torch.zeros(1, 64, 1024, device=dev)
def chain():
    model.bvrnn.decode(codes, h0)
for s in cands:                      # per-stream workspaces + graphs
    with torch.cuda.stream(s):
        chain(); chain()
torch.cuda.synchronize()
def timed(ss, reps=4):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            for s in ss:
                with torch.cuda.stream(s):
                    chain()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
alone = min(timed([s]) for s in cands[:3])
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES','default')} pg={os.environ.get('BVC_FORCE_PG','0')} alone {alone*1e3:.2f} ms")
for i in range(N):
    print(" ".join(f"{timed([cands[i], cands[j]]) / alone:4.2f}" if j > i else "  . " for j in range(N)))
