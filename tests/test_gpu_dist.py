"""The RCCL leg of the multi-GPU path on the one GPU the box has: a 1-rank "nccl" process group (BVC_FORCE_PG=1),
the per-step all_gather of bench.py on measured-concurrent streams, and the sharded facade.  Runs in a child
process so that the communicator does not outlive the test."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys, tempfile, torch
sys.path.insert(0, os.environ["BVC_ROOT"])
from bvcodec import BVRNNCodecModel, config, dist as bdist, synth
rank, world, device = bdist.init_from_env()
assert torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl" and world == 1
conf = config.load_config(config.DEFAULT_CONFIG)
d = tempfile.mkdtemp()
p1, p2 = synth.write_checkpoints(conf, d, seed=1234)
model = BVRNNCodecModel(config.DEFAULT_CONFIG, p1, p2).to(device)
B, L = 16, 22050
x = synth.synthetic_speech(B, L, seed=3, kind="speech").to(device)
codes0 = model.encode(x, 3000); wav0 = model.decode(codes0, L)
torch.cuda.synchronize()
streams = bdist.concurrent_stream_sets(2, device)[0]
outs = [torch.empty(world * B, L, device=device) for _ in streams]
for sched in ("persistent", "auto"):
    # 'persistent': every call a persistent launch, each of which must wait for the collectives issued before it on the
    # OTHER stream as well (bvc_flow_fence through bdist.fence_collective: an RCCL kernel that waits for peers holds compute
    # units the launch needs); 'auto': the default, which takes the layer kernels once the two streams overlap
    model.set_recurrence(sched)
    for o in outs:
        o.zero_()
    for k in range(6):                              # three gathered steps per stream
        with torch.cuda.stream(streams[k % 2]):
            codes = model.encode(x, 3000)
            wav = model.decode(codes, L)
            torch.distributed.all_gather_into_tensor(outs[k % 2], wav)
            bdist.fence_collective(device)
    for st in streams:
        torch.cuda.current_stream(device).wait_stream(st)
    torch.cuda.synchronize()
    model.check_status()
    for o in outs:                                  # (whichever kernels 'auto' picked: one order of summation per output)
        assert torch.equal(o, wav0), "gathered waveform differs from the local one"
model.set_recurrence("auto")
c2, w2 = bdist.codec_sharded(model, x, 3000, gather=True)
assert torch.equal(c2, codes0) and torch.equal(w2, wav0)
model.check_status()
torch.distributed.barrier()
torch.distributed.destroy_process_group()
print("DIST-OK")
'''


def test_rccl_one_rank_gather_on_picked_streams():
    env = dict(os.environ, BVC_FORCE_PG="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29600 + os.getpid() % 300), BVC_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0 and "DIST-OK" in r.stdout, r.stdout[-2000:] + "\n" + r.stderr[-4000:]


def test_bench_target_workload_leg_runs_under_a_process_group():
    """bench.py with a process group (what the driver's `--gpus N` runs get): the configs[3] leg - 64 x 10 s per rank, bitrates
    1.5 / 3 / 6 kbit/s, the per-step RCCL all-gather inside the timed region - runs on every rank and is reported by rank 0."""
    import json
    env = dict(os.environ, BVC_FORCE_PG="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29900 + os.getpid() % 90), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--multi-streams", "0",
                        "--no-cpu-baseline", "--no-roofline", "--legs", "target"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["config"]["gather"].startswith("rccl") and line["gather_ms"] > 0
    tw = line["target_workload"]
    assert tw["n_gpus"] == 1 and tw["gather_ms"] > 0 and tw["frames_per_utterance"] == 861
    assert set(tw["by_bitrate"]) == {"1500", "3000", "6000"} and all(v["value"] > 1000 for v in tw["by_bitrate"].values())
    assert tw["parity"]["max_first_divergence_margin"] < 1e-5 and tw["parity"]["waveform_rms_error"] < 1e-4
    assert len(tw["rank_ms_per_step"]["per_rank"]) == 1
