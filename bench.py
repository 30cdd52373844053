#!/usr/bin/env python3
"""Benchmark of the BVRNNCodecModel encode+decode hot path on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: ``encode(x, 3000)`` then
``decode(codes, L)`` on BASELINE.json configs[1] - 64 synthetic 5 s utterances at 22.05 kHz,
config_varBitRate @ 3 kbit/s (35 bits/frame) - PER GPU (weak scaling: utterances are independent,
each rank runs its own batch with no data-path collective; one RCCL all-gather of the decoded
waveforms per step is the "final gather" of the north star).  Inputs are resident in HBM before the
timed region.  Rank 0 prints ONE JSON line: metric/value (audio-seconds coded per wall-second,
whole job), plus

  roofline     - the dominant kernel family, timed in situ with hipEvent pairs around sampled
                 launches inside the real schedule (bvc_probe_*), algorithmic FLOPs per launch from
                 SURVEY.md 8(d), against the fp32 MFMA peak of MI355X_MICROARCH.md
  cpu_baseline - the CPU oracle (oracle/, PyTorch-CPU port of the reference op sequence) timed on
                 the host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time

# The four-stream schedule relies on the runtime's default of 4 hardware queues per process (one per stream): 3, 5 or 6
# queues cost 20-60 % of the throughput on this pool.  Pin the default unless the
# caller chose otherwise; it must be in the environment before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bvcodec import BVRNNCodecModel, _abi, config, dist as bdist, synth   # noqa: E402

FS = 22050
BATCH = 64
SECONDS = 5.0
BITRATE = 3000
PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: dense fp32 matrix = vector peak
PROBE_NAMES = {1: "gemm_skinny_kernel<1,1,8,2,1> / <1,1,12,1,1> (recurrent BVRNN layer)", 2: "gemm_skinny_kernel<3,2,12,1,1,gate-interleaved> (GRU cell)",
               3: "amp_pair_kernel / conv_mfma_kernel (BigVGAN convs)", 4: "gemm_batched_(lds_)kernel (phi_x / phi_z and the decoder's phi_z products over all frames)",
               5: "stft_logmel_kernel", 6: "conv_post_kernel"}


def flops_per_step(conf, B, T):
    """Algorithmic FLOPs (2*MAC) of one encode+decode step, per kernel family."""
    H, Z, X = conf["h_dim"], conf["z_dim"], conf["num_mels"]
    enc_lin = 2 * H * H + H * H + H * Z + (Z * H + 2 * H * H) + (2 * H * H + 2 * H * H + H * X) + (X * H + 2 * H * H)
    dec_lin = (Z * H + 2 * H * H) + (2 * H * H + 2 * H * H + H * X) + (X * H + 2 * H * H)
    gru = 3 * H * (2 * H) + 3 * H * H
    phix = X * H + 2 * H * H
    v = conf["vocoder_config"]
    ch, rate, voc = v["upsample_initial_channel"], 1, conf["num_mels"] * v["upsample_initial_channel"] * 7
    for u, k in zip(v["upsample_rates"], v["upsample_kernel_sizes"]):
        rate *= u
        voc += rate * ch * (ch // 2) * (k // u)                      # transposed conv
        ch //= 2
        voc += rate * ch * ch * sum(v["resblock_kernel_sizes"]) * 6  # 3 AMP blocks x 3 x 2 convs
    post = rate * ch * 7
    BT = B * T
    return {
        1: (2.0 * BT * (enc_lin + dec_lin), T * (13 + 10)),
        2: (2.0 * BT * 2 * gru, 2 * T),
        3: (2.0 * BT * voc, 1 + len(v["upsample_rates"]) * (1 + 9)),      # conv_pre + per stage: ConvT + 9 fused AMP iterations
        4: (2.0 * BT * phix, 3),
        5: (2.0 * BT * 5 * 512 * 9 * 1.0, 1),
        6: (2.0 * BT * post, 1),
    }


def pmc_traffic_bytes(kernel_prefix):
    """HBM-side bytes per launch of a kernel family from the committed rocprofv3 --pmc pass
    (FETCH_SIZE x2 per the gfx950 correction of MI355X_MICROARCH.md + WRITE_SIZE), launch-weighted.
    PMC collection serialises every dispatch, so it is a separate pass (tools/pmc_probe.py), not part
    of this run; None if the summary is absent."""
    path = os.path.join(ROOT, "profiles", "r01_f_pmc_hbm_traffic_per_launch.csv")
    if not os.path.exists(path):
        return None
    import csv
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if kernel_prefix in r["kernel"]:
            k = int(r["launches"])
            tot += k * (2.0 * float(r["FETCH_SIZE_KB_raw_avg"]) + float(r["WRITE_SIZE_KB_avg"])) * 1024.0
            n += k
    return round(tot / n) if n else None


def cpu_baseline(conf):
    """Oracle (PyTorch-CPU port of the reference op sequence) on a bounded sample: 32 x 5 s, half the GPU
    batch (about 10 s of CPU work on the box's 16 cores)."""
    from oracle import codec as ocodec
    threads = min(16, os.cpu_count() or 1)      # the 1-GPU box's CPU share is 16 cores
    torch.set_num_threads(threads)
    oc = ocodec.OracleCodec(conf, synth.bvrnn_state_dict(conf, 1234), synth.generator_state_dict(conf, 1235))
    b = 32
    x = synth.synthetic_speech(b, int(FS * SECONDS), seed=0, kind="noise")
    oc.forward(x[:1, :FS], BITRATE)                                  # warm-up (thread pools, mkldnn)
    t0 = time.time()
    oc.forward(x, BITRATE)
    dt = time.time() - t0
    return {"value": round(b * SECONDS / dt, 3), "unit": "audio-seconds/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{b} x {SECONDS:g} s utterances, encode+decode @ {BITRATE} bit/s, "
                                      f"oracle (PyTorch-CPU eager fp32), {dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help="utterances per GPU")
    ap.add_argument("--seconds", type=float, default=SECONDS)
    ap.add_argument("--streams", type=int, default=4,
                    help="HIP streams the K steps are issued on round-robin (independent batches overlap)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    a = ap.parse_args()

    # RCCL prints a version banner on STDOUT when its communicator is created; this program's stdout is one JSON line, so
    # the process group is set up (and its first collective run) with fd 1 pointing at stderr
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        rank, world, device = bdist.init_from_env()
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.barrier()
            torch.cuda.synchronize(device)
    finally:
        sys.stdout.flush()
        os.dup2(saved_fd, 1)
        os.close(saved_fd)
    if world != a.gpus and rank == 0:
        print(f"warning: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    assert device.type == "cuda", "bench.py needs an MI355X"
    conf = config.load_config(config.DEFAULT_CONFIG)
    ckdir = tempfile.mkdtemp(prefix=f"bvc_bench_r{rank}_")
    p1, p2 = synth.write_checkpoints(conf, ckdir, seed=1234)
    model = BVRNNCodecModel(config.DEFAULT_CONFIG, p1, p2).to(device)

    L = int(FS * a.seconds)
    B = a.batch
    x = synth.synthetic_speech(B, L, seed=rank, kind="noise").to(device)      # resident before timing
    use_pg = torch.distributed.is_available() and torch.distributed.is_initialized()
    nstreams = max(1, a.streams)
    gathered = ([torch.empty(world * B, L, device=device) for _ in range(nstreams)]
                if (use_pg and not a.no_gather) else None)

    def local_step():
        codes = model.encode(x, BITRATE)
        return codes, model.decode(codes, L)

    def step(slot=0):
        codes, wav = local_step()
        if gathered is not None:                  # the "final gather" of the north star: one RCCL collective
            torch.distributed.all_gather_into_tensor(gathered[slot], wav)
        return codes, wav

    # Steps are independent batches: issue them round-robin on `--streams` HIP streams so that the
    # latency-bound recurrent chain of one batch overlaps the next batch's work (every step still
    # runs the full encode -> decode; all K steps complete inside the timed bracket).
    # ... on streams that really are concurrent: which HIP streams share a hardware queue is not visible through the API
    # and shifts when an RCCL communicator exists, so the set is measured (bvcodec.dist.concurrent_streams, ~0.3 s)
    try:
        stream_sets = bdist.concurrent_stream_sets(nstreams, device) if nstreams > 1 else [[torch.cuda.Stream(device)]]
    except Exception as e:                      # never let the placement probe take the benchmark down
        print(f"warning: stream placement probe failed ({e!r}); using the first {nstreams} streams", file=sys.stderr)
        stream_sets = [[torch.cuda.Stream(device) for _ in range(nstreams)]]
    streams = stream_sets[0]

    def run(n):
        last = None
        for k in range(n):
            st = streams[k % len(streams)]
            with torch.cuda.stream(st):
                last = step(k % len(streams))
        for st in streams:
            torch.cuda.current_stream(device).wait_stream(st)
        return last

    if len(stream_sets) > 1:
        # the pair test uses tiny kernels; settle between the candidate sets with the real workload (untimed, before warm-up)
        # (local steps only: the number of candidate sets may differ between ranks, so no collective in here)
        def run_local(cand, n):
            for k in range(n):
                with torch.cuda.stream(cand[k % len(cand)]):
                    local_step()
            torch.cuda.synchronize(device)

        best = None
        for cand in stream_sets:
            run_local(cand, len(cand))
            t_try = time.perf_counter()
            run_local(cand, 2 * len(cand))
            t_try = time.perf_counter() - t_try
            if best is None or t_try < best[0]:
                best = (t_try, cand)
        streams = best[1]
    run(max(a.warmup, len(streams)) if a.warmup else 0)
    if use_pg:
        torch.distributed.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    codes, wav = run(a.steps)
    torch.cuda.synchronize(device)
    if use_pg:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if use_pg:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert torch.isfinite(wav).all() and codes.shape[2] == conf["z_dim"]

    T = codes.shape[1]
    out = {
        "metric": "audio-seconds coded per wall-second (encode+decode), 22.05 kHz @ 3 kbit/s",
        "value": round(world * B * a.seconds * a.steps / elapsed, 2),
        "unit": "audio-seconds/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1e3 * elapsed / a.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: batch {B} x {a.seconds:g} s utterances per GPU, "
                               f"config_varBitRate @ {BITRATE} bit/s (35 bits/frame), full encode -> BigVGAN decode",
                   "frames_per_utterance": T, "weights": "seeded synthetic (checkpoints are LFS pointers)",
                   "gather": "rccl all_gather of decoded waveforms" if gathered is not None else "none",
                   "streams": len(streams)},
    }

    if rank == 0 and len(streams) > 1:
        # for reference: the same K steps issued back to back on ONE stream (latency-oriented number)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        for _ in range(a.steps):
            local_step()
        torch.cuda.synchronize(device)
        dt1 = time.perf_counter() - t1
        out["single_stream"] = {"value": round(B * a.seconds * a.steps / dt1, 2), "ms_per_step": round(1e3 * dt1 / a.steps, 3),
                                "note": "one rank, one stream, no gather"}

    if rank == 0 and not a.no_roofline:
        lib = _abi.load()
        fam = flops_per_step(conf, B, T)
        rows = {}

        def kread(lo, hi):
            mean, mn, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int32()
            _abi.check(lib.bvc_kprobe_read(lo, hi, ctypes.byref(mean), ctypes.byref(mn), ctypes.byref(n)))
            return mean.value, n.value

        # recurrent kernels are replayed from a hipGraph: in-kernel wall-clock stamps (bvc_kprobe_*)
        _abi.check(lib.bvc_kprobe_enable(1))
        c_ = model.encode(x, BITRATE)
        e_lin, e_nl = kread(0, 13)
        e_gru, e_ng = kread(13, 14)
        model.decode(c_, L)
        d_lin, d_nl = kread(0, 7)
        d_gru, d_ng = kread(7, 8)
        _abi.check(lib.bvc_kprobe_enable(0))
        fam[1] = (fam[1][0] - 2.0 * B * T * (conf["z_dim"] * conf["h_dim"] + 2 * conf["h_dim"] ** 2), T * (13 + 7))
        fam[4] = (fam[4][0] + 2.0 * B * T * (conf["z_dim"] * conf["h_dim"] + 2 * conf["h_dim"] ** 2), 6)
        if os.environ.get("BVC_NO_PRECOMP", "0") != "1" and os.environ.get("BVC_SIDE_BRANCH", "0") != "1":
            # decode: the phi_z halves of dec.0 (H x H) and of the GRU input product (3H x H) are batched over all frames
            hh = 2.0 * B * T * conf["h_dim"] ** 2
            fam[1] = (fam[1][0] - hh, fam[1][1])
            fam[2] = (fam[2][0] - 3 * hh, fam[2][1])
            fam[4] = (fam[4][0] + 4 * hh, 8)
        for kind, mean, n in ((1, (e_lin * e_nl + d_lin * d_nl) / max(1, e_nl + d_nl), e_nl + d_nl),
                              (2, (e_gru * e_ng + d_gru * d_ng) / max(1, e_ng + d_ng), e_ng + d_ng)):
            fl, launches = fam[kind]
            rows[kind] = {"kernel": PROBE_NAMES[kind], "launches_per_step": launches, "sampled": n,
                          "mean_us": round(mean, 3), "total_ms": round(mean * launches / 1e3, 3),
                          "tflops": round(fl / launches / (mean * 1e-6) / 1e12, 3) if mean else 0.0,
                          "timer": "in-kernel wall_clock64 (first workgroup start -> last workgroup end)"}
        for kind in (3, 4, 5, 6):               # eagerly launched kernels: hipEvent pairs around every launch
            _abi.check(lib.bvc_probe_begin(kind, 1, 4096))
            local_step()                        # rank 0 only: no collective here
            mean, mn, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int32()
            _abi.check(lib.bvc_probe_end(ctypes.byref(mean), ctypes.byref(mn), ctypes.byref(n)))
            fl, launches = fam[kind]
            rows[kind] = {"kernel": PROBE_NAMES[kind], "launches_per_step": launches, "sampled": n.value,
                          "mean_us": round(mean.value, 3), "total_ms": round(mean.value * launches / 1e3, 3),
                          "tflops": round(fl / launches / (mean.value * 1e-6) / 1e12, 3) if mean.value else 0.0,
                          "timer": "hipEvent pair on the launch stream"}
        dom = max(rows, key=lambda k: rows[k]["total_ms"])
        r = rows[dom]
        out["roofline"] = {"bound": "mfma", "achieved": r["tflops"], "peak": PEAK_FP32_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": round(r["tflops"] / PEAK_FP32_MFMA_TFLOPS, 4),
                           "traffic": pmc_traffic_bytes("gemm_skinny_kernel<1, 1,") if dom == 1 else None,
                           "traffic_unit": "bytes per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate pass; "
                                           "algorithmic: 4.5e6 for a 1024x1024 layer at B=64)",
                           "kernel": r["kernel"], "mean_launch_us": r["mean_us"],
                           "launches_per_step": r["launches_per_step"],
                           "timer": r["timer"],
                           "rocprof_avg_us": "5.2 (rocprofv3 --kernel-trace, dispatch-inclusive, --streams 1: profiles/r01_p_kernel_stats_final_1stream.csv)",
                           "note": "fp32-in/fp32-acc MFMA (v_mfma_f32_16x16x4_f32); algorithmic FLOPs per launch / "
                                   "launch duration measured in situ inside the real schedule.  The layer is not MFMA-limited: "
                                   "each 16x16 output tile pulls 128 KB of operands through its CU's L1, and chains of this "
                                   "layer shape saturate the chip at 0.37 layers/us = 50 TFLOP/s whatever the tiling "
                                   "(profiles/r01_concurrency_microbench.txt)"}
        # SURVEY.md 8(d) also asks for the whole path against the fp32 peak: algorithmic FLOPs of a step (all families)
        # over the measured step time of the timed region (the kernels of the streams overlap, so this is not the
        # sum of the per-launch figures above)
        step_flops = sum(f for f, _ in flops_per_step(conf, B, T).values())
        whole = step_flops / (elapsed / a.steps) / 1e12
        out["roofline"]["whole_path"] = {"achieved": round(whole, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                         "frac": round(whole / PEAK_FP32_MFMA_TFLOPS, 4),
                                         "gflop_per_audio_second": round(step_flops / (B * a.seconds) / 1e9, 3)}
        out["kernel_families"] = rows
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(conf)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_pg:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
