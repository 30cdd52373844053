#!/usr/bin/env python3
"""Benchmark of the BVRNNCodecModel encode+decode hot path on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over ONE batch: ``encode(x, 3000)`` then ``decode(codes, L)`` on
BASELINE.json configs[1] - 64 synthetic 5 s utterances at 22.05 kHz, config_varBitRate @ 3 kbit/s (35
bits/frame) - PER GPU (weak scaling: utterances are independent, each rank runs its own batch with no
data-path collective; one RCCL all-gather of the decoded waveforms per step is the "final gather" of the
north star).  Inputs are resident in HBM before the timed region; the K steps are issued back to back on
one stream.  Rank 0 prints ONE JSON line: metric/value (audio-seconds coded per wall-second, whole job), plus

  roofline      - the dominant kernel (the persistent BVRNN recurrence, two launches per step) timed with HIP
                  event pairs on its launch stream inside the real schedule (bvc_probe_*), algorithmic FLOPs per
                  launch from SURVEY.md 8(d), against the fp32 MFMA peak of MI355X_MICROARCH.md;
                  `rocprof_avg_us` / `frac_rocprof` quote the committed rocprofv3 --kernel-trace summary
  parity        - after the timed region: the timed batch's codes against the CPU oracle on sampled
                  utterances (differing bits, margin of the first divergence) and the decoded waveform's RMS error
  multi_stream  - the same K steps issued round-robin on four HIP streams (4 x 64 utterances in flight); the library's
                  default recurrence schedule ("auto") switches to the launch-per-layer kernels by itself there
  gather_ms     - mean duration of the per-step RCCL all-gather (HIP event pair on its stream; 0 without a process group)
  forward_fused - forward(x, bitrate) with ONE recurrence (bvc_forward; clearly NOT the headline, which stays encode() + decode())
  target_workload / encode_only / streaming - BASELINE configs[3]'s per-GPU shard (64 x 10 s, the north-star target's
                  utterance length; bitrates 1.5 / 3 / 6 kbit/s; with --gpus N > 1 EVERY rank runs its shard, the per-step
                  all-gather included, max over ranks), configs[2] (encode only) and configs[4] (256 streams x 20 ms hops,
                  one library call per hop: p50 / p99), each timed after the headline with its own parity
                  spot check (the last two at N = 1 only)
  cpu_baseline  - the CPU oracle (oracle/, PyTorch-CPU port of the reference op sequence) timed on the host
                  cores on a bounded sample of the same workload (rank 0, N=1 only): the benchmark's batch of 64, one
                  second per utterance, 3 warm-ups, median of 5, at 8 threads and at all usable cores.

Other workloads of BASELINE.json: ``--seconds 10`` (the north-star target's utterance length, and with
``--bitrate 1500|3000|6000`` one GPU's shard of configs[3]), ``--mode encode`` (configs[2]: front-end + coder only).
"""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time

# The multi-stream schedule relies on the runtime's default of 4 hardware queues per process (one per stream): 3, 5 or 6
# queues cost 20-60 % of its throughput on this pool.  Pin the default unless the caller chose otherwise; it must be in the
# environment before the HIP runtime initialises.
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
PROBE_NAMES = {1: "bvrnn_flow_kernel<8,encode> / <8,decode> (persistent BVRNN recurrence: all frames of one call in one launch)",
               3: "amp_pair_kernel / conv_mfma_kernel (BigVGAN convs)",
               4: "gemm_batched_(lds_)kernel (phi_x / phi_z and the frame-independent halves of enc.0, dec.0 and the GRU input gates, all frames)",
               5: "stft_logmel_kernel", 6: "conv_post_kernel"}
def _latest(name):
    for tag in ("r04", "r03", "r02"):
        p = os.path.join(ROOT, "profiles", f"{tag}_{name}")
        if os.path.exists(p):
            return p
    return os.path.join(ROOT, "profiles", f"r04_{name}")


ROCPROF_SUMMARY = _latest("kernel_stats_default.csv")
PMC_SUMMARY = _latest("pmc_per_launch.csv")


def flops_per_step(conf, B, T, mode, fold=True):
    """Algorithmic FLOPs (2*MAC) of one step per kernel family -> {family: (flops, launches)}.
    fold: the persistent decode kernel runs phi_x.0(norm(dec.6(u))) as one folded H x H layer and dec.6 itself as a tenth batched
    GEMM (library default).  Counted are the REFERENCE's products: phi_x.0 as X*H in the recurrence (not the H*H the
    folded layer really multiplies), dec.6 where it now runs (batched)."""
    H, Z, X = conf["h_dim"], conf["z_dim"], conf["num_mels"]
    # recurrence, per frame and utterance.  encode: enc.0 (h half), enc.2, enc.4, phi_z x3, dec.0 (both halves), dec.2/4/6,
    # phi_x x3, GRU (W_ih 3Hx2H, W_hh 3HxH).  decode: dec.0 (h half), dec.2/4/6, phi_x x3, GRU (W_ih[:, :H], W_hh)
    rec_enc = (H * H + H * H + H * Z) + (Z * H + 2 * H * H) + (2 * H * H + 2 * H * H + H * X) + (X * H + 2 * H * H) + 9 * H * H
    rec_dec = (H * H + 2 * H * H + H * X) + (X * H + 2 * H * H) + 6 * H * H
    # batched over all frames.  encode: phi_x + enc.0[:, :H]; decode: phi_z + dec.0[:, :H] + W_ih[:, H:]
    bat_enc = (X * H + 2 * H * H) + H * H
    bat_dec = (Z * H + 2 * H * H) + H * H + 3 * H * H
    if fold:
        rec_dec -= H * X
        bat_dec += H * X
    v = conf["vocoder_config"]
    ch, rate, voc = v["upsample_initial_channel"], 1, conf["num_mels"] * v["upsample_initial_channel"] * 7
    for u, k in zip(v["upsample_rates"], v["upsample_kernel_sizes"]):
        rate *= u
        voc += rate * ch * (ch // 2) * (k // u)                      # transposed conv
        ch //= 2
        voc += rate * ch * ch * sum(v["resblock_kernel_sizes"]) * 6  # 3 AMP blocks x 3 x 2 convs
    post = rate * ch * 7
    BT = B * T
    full = mode == "codec"
    return {
        1: (2.0 * BT * (rec_enc + (rec_dec if full else 0)), 2 if full else 1),
        3: (2.0 * BT * voc if full else 0.0, 1 + len(v["upsample_rates"]) * (1 + 9)),   # conv_pre + per stage: ConvT + 9 fused AMP iterations
        4: (2.0 * BT * (bat_enc + (bat_dec if full else 0)), (10 if fold else 9) if full else 4),
        5: (2.0 * BT * 5 * 512 * 9 * 1.0, 1),
        6: (2.0 * BT * post if full else 0.0, 1),
    }


def rocprof_avg_us(kernel_substr):
    """Launch-weighted average duration of a kernel family in the committed rocprofv3 --kernel-trace --stats summary of
    the default command (profiles/, regenerated by tools/profile_round.sh); None if absent."""
    if not os.path.exists(ROCPROF_SUMMARY):
        return None
    import csv
    tot, n = 0.0, 0
    for r in csv.DictReader(open(ROCPROF_SUMMARY)):
        if kernel_substr in r.get("Name", ""):
            tot += float(r["TotalDurationNs"]) / 1e3
            n += int(r["Calls"])
    return round(tot / n, 3) if n else None


def pmc_l2_to_cu_bytes(kernel_substr):
    """Bytes that came out of the L2s into the compute units per launch (TCP_TCC_READ_REQ_sum x 128 B) from the same committed PMC summary,
    launch-weighted; None if absent.  This - not HBM and not the matrix pipe - is what the recurrence kernel leans on."""
    if not os.path.exists(PMC_SUMMARY):
        return None
    import csv
    tot, n = 0.0, 0
    for r in csv.DictReader(open(PMC_SUMMARY)):
        if kernel_substr in r["kernel"] and r.get("TCP_TCC_READ_REQ_sum_avg"):
            k = int(r["launches"])
            tot += k * float(r["TCP_TCC_READ_REQ_sum_avg"]) * 128.0
            n += k
    return round(tot / n) if n else None


def pmc_traffic_bytes(kernel_substr):
    """HBM-side bytes per launch from the committed rocprofv3 --pmc pass (FETCH_SIZE x2 per the gfx950 correction of
    MI355X_MICROARCH.md + WRITE_SIZE), launch-weighted.  PMC collection serialises every dispatch, so it is a separate pass
    (tools/pmc_probe.py), not part of this run; None if the summary is absent."""
    if not os.path.exists(PMC_SUMMARY):
        return None
    import csv
    tot, n = 0.0, 0
    for r in csv.DictReader(open(PMC_SUMMARY)):
        if kernel_substr in r["kernel"]:
            k = int(r["launches"])
            tot += k * (2.0 * float(r["FETCH_SIZE_KB_raw_avg"]) + float(r["WRITE_SIZE_KB_avg"])) * 1024.0
            n += k
    return round(tot / n) if n else None


def oracle_leg(conf, model, x, codes, wav, L, bitrate, mode, with_baseline):
    """The CPU oracle as checker and as reference point (rank 0).  parity: the oracle encodes sampled utterances of the TIMED
    batch; the codes the GPU produced in the timed region are compared bit by bit; the oracle then decodes the GPU's codes
    and the waveform is compared.  cpu_baseline: the oracle timed on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from parity_stats import divergence_stats
    oc = get_oracle(conf)
    out = {}
    idx = [0, x.shape[0] // 3, (2 * x.shape[0]) // 3, x.shape[0] - 1]
    xs = x[idx].cpu()
    r = oc.encode(xs, bitrate, full=True)
    nb = int(model.bits_per_frame(bitrate)) if conf["var_bit"] else conf["z_dim"]
    st = divergence_stats(codes[idx].cpu(), r["codes"], r["prob"], min(nb, conf["z_dim"]))
    par = {"checker": "oracle/ (CPU restatement of the reference, pinned to its goldens)", "utterances_checked": idx,
           "code_bits_checked": len(idx) * codes.shape[1] * min(nb, conf["z_dim"]),
           "code_bits_differing": st["mismatching_bits_total"], "utterances_diverged": st["diverged_utterances"],
           "max_first_divergence_margin": st["max_first_divergence_margin"],
           "bits_within_1e-6_of_a_tie": st["bits_within_1e-6_of_a_tie"]}
    if mode == "codec":
        ref_wav = oc.decode(codes[idx[:2]].cpu(), L)
        par["waveform_rms_error"] = float((wav[idx[:2]].cpu() - ref_wav).pow(2).mean().sqrt())
        par["waveform_rms"] = float(ref_wav.pow(2).mean().sqrt())
        par["waveform_utterances_checked"] = idx[:2]
    out["parity"] = par
    if with_baseline:
        out["cpu_baseline"] = cpu_baseline(oc, L, bitrate, mode)
    return out


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """Cores this process may really use: the affinity mask, cut to the cgroup's CPU quota where there is one (a GPU box hands
    one GPU's job a share of the host - 16 cores - while every core stays visible; more threads than that only queue)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return min(n, int(os.environ.get("BVC_BENCH_MAX_THREADS", "16")))


def cpu_baseline(oc, L, bitrate, mode):
    """the oracle on a bounded sample of the workload, 3 warm-ups, median of 5, at 8 threads (the
    reference was timed at 8 in the build container) and at all the cores this process may use."""
    b, secs = BATCH, 1.0            # the benchmark's batch (the CPU vocoder's cost depends on it), one second each
    xb = synth.synthetic_speech(b, int(FS * secs), seed=0, kind="noise")
    fn = (lambda t: oc.forward(t, bitrate)) if mode == "codec" else (lambda t: oc.encode(t, bitrate))
    allc = usable_cores()
    runs = {}
    t_all = time.time()
    for threads in sorted({min(8, allc), allc}):
        torch.set_num_threads(threads)
        for _ in range(3):
            fn(xb[:2])                                                   # warm-ups (thread pools, oneDNN primitives)
        ts = []
        for _ in range(5):
            t0 = time.time()
            fn(xb)
            ts.append(time.time() - t0)
        ts.sort()
        runs[threads] = {"value": round(b * secs / ts[2], 3), "median_s": round(ts[2], 3), "min_s": round(ts[0], 3), "max_s": round(ts[-1], 3)}
    best = max(runs, key=lambda k: runs[k]["value"])
    return {"value": runs[best]["value"], "unit": "audio-seconds/s", "cores": best, "kind": "port",
            "sample": f"{b} x {secs:g} s utterances, {'encode+decode' if mode == 'codec' else 'encode only'} @ {bitrate:g} bit/s, oracle "
                      f"(PyTorch-CPU eager fp32), 3 warm-ups + median of 5 per thread count, {time.time() - t_all:.0f} s wall in all",
            "by_threads": {str(k): v for k, v in runs.items()}, "cpu_model": cpu_model_name(), "cores_usable": allc,
            "cores_visible": os.cpu_count()}


_ORACLE = {}
_T0 = time.time()


def note(msg):
    """Progress on stderr (stdout carries the one JSON line): a long run must not look hung."""
    print(f"[bench {time.time() - _T0:6.1f} s] {msg}", file=sys.stderr, flush=True)



def get_oracle(conf):
    if "oc" not in _ORACLE:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle import codec as ocodec
        torch.set_num_threads(usable_cores())
        _ORACLE["oc"] = ocodec.OracleCodec(conf, synth.bvrnn_state_dict(conf, 1234), synth.generator_state_dict(conf, 1235))
    return _ORACLE["oc"]


def spot_check(conf, model, x_rows, codes_rows, wav_rows, n_wav, bitrate, frames=None):
    """Oracle check of a few rows of a leg's output: code bits (free-running; margin of the first divergence, if any) and the
    waveform of the oracle decoding the GPU's codes.  x_rows (n, L) / codes_rows (n, F, z) / wav_rows (n, >= n_wav) on any device."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from parity_stats import divergence_stats
    oc = get_oracle(conf)
    r = oc.encode(x_rows.cpu(), bitrate, full=True)
    F = codes_rows.shape[1] if frames is None else frames
    nb = min(int(model.bits_per_frame(bitrate)) if conf["var_bit"] else conf["z_dim"], conf["z_dim"])
    st = divergence_stats(codes_rows.cpu()[:, :F], r["codes"][:, :F], r["prob"][:, :F], nb)
    par = {"code_bits_checked": int(codes_rows.shape[0] * F * nb), "code_bits_differing": st["mismatching_bits_total"],
           "max_first_divergence_margin": st["max_first_divergence_margin"]}
    if wav_rows is not None:
        ref = oc.decode(codes_rows.cpu()[:, :F], n_wav)
        par["waveform_rms_error"] = float((wav_rows.cpu()[:, :n_wav] - ref).pow(2).mean().sqrt())
        par["waveform_rms"] = float(ref.pow(2).mean().sqrt())
    return par


def time_steps(fn, n, device):
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    last = None
    for _ in range(n):
        last = fn()
    torch.cuda.synchronize(device)
    return time.perf_counter() - t0, last


def leg_target_workload(conf, model, device, B, bitrate, with_parity, steps=5, world=1, use_pg=False, rank=0, sweep=(1500.0, 6000.0)):
    """The workload the north-star target and its scaling are quoted on: BASELINE configs[3], 512 x 10 s sharded over 8 GPUs = 64 x 10 s
    per GPU, bitrates {1.5, 3, 6} kbit/s.  With a process group EVERY rank runs its shard and the per-step RCCL all-gather of the
    decoded waveforms is inside the timed region (barrier + synchronize on both sides, max over ranks), so that a `--gpus N` run
    lands on this workload too; rank 0 reports."""
    L = int(FS * 10.0)
    x = synth.synthetic_speech(B, L, seed=1000 + rank, kind="noise").to(device)
    gathered = torch.empty(world * B, L, device=device) if use_pg else None
    gather_ev = []

    def make_step(br):
        def step():
            codes = model.encode(x, br)
            wav = model.decode(codes, L)
            if gathered is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                torch.distributed.all_gather_into_tensor(gathered, wav)
                e1.record()
                gather_ev.append((e0, e1))
                bdist.fence_collective(device)
            return codes, wav
        return step

    def timed_leg(br, n):
        step = make_step(br)
        step()
        gather_ev.clear()
        if use_pg:
            torch.distributed.barrier()
        dt, last = time_steps(step, n, device)
        if use_pg:
            torch.distributed.barrier()
        g_ms = (sum(e0.elapsed_time(e1) for e0, e1 in gather_ev) / len(gather_ev)) if gather_ev else 0.0
        rank_ms = [round(1e3 * dt / n, 3)]
        if use_pg:
            tl = torch.tensor([dt], device=device, dtype=torch.float64)
            tall = [torch.zeros_like(tl) for _ in range(world)]
            torch.distributed.all_gather(tall, tl)
            rank_ms = [round(1e3 * float(t.item()) / n, 3) for t in tall]
            dt = max(float(t.item()) for t in tall)
        model.check_status()
        return dt, last, g_ms, rank_ms

    dt, (codes, wav), g_ms, rank_ms = timed_leg(bitrate, steps)
    out = {"workload": f"BASELINE configs[3]: batch {B} x 10 s per GPU ({world * B} x 10 s over {world} GPU{'s' if world > 1 else ''}) @ {bitrate:g} bit/s, "
                       f"full encode -> BigVGAN decode, one batch at a time{', RCCL all-gather of the decoded waveforms per step' if use_pg else ''}",
           "value": round(world * B * 10.0 * steps / dt, 2), "unit": "audio-seconds/s", "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps,
           "n_gpus": world, "frames_per_utterance": int(codes.shape[1]), "x_real_time": round(world * B * 10.0 * steps / dt, 1),
           "gather_ms": round(g_ms, 3), "rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms), "per_rank": rank_ms}}
    by = {f"{bitrate:g}": {"value": out["value"], "ms_per_step": out["ms_per_step"]}}
    for br in sweep:                              # the other two bitrates of configs[3] (17 and 64 active bits per frame)
        if br == bitrate:
            continue
        n2 = max(2, steps // 2)
        dt2, _, g2, _ = timed_leg(br, n2)
        by[f"{br:g}"] = {"value": round(world * B * 10.0 * n2 / dt2, 2), "ms_per_step": round(1e3 * dt2 / n2, 3), "gather_ms": round(g2, 3)}
    out["by_bitrate"] = by
    if with_parity and rank == 0:
        out["parity"] = dict(spot_check(conf, model, x[:1], codes[:1], wav[:1], L, bitrate), utterances_checked=[0])
    del gathered
    return out


def leg_forward_fused(conf, model, device, x, bitrate, with_parity, steps=5):
    """NOT the headline and not a BASELINE config: BVRNNCodecModel.forward(x, bitrate) (bvrnn_codec_model.py:73-76, example.py:20) on the
    benchmark's batch through bvc_forward - the encoder's frame loop already runs the decoder on every frame, so its outputs go to
    the vocoder and the second recurrence (and the all-frame phi_z GEMMs) of decode(encode(x)) are not run again."""
    B, L = x.shape
    model.forward_fused(x, bitrate)
    dt, wav = time_steps(lambda: model.forward_fused(x, bitrate), steps, device)
    model.check_status()
    out = {"workload": f"forward(x, {bitrate:g}) on batch {B} x {L / FS:g} s through bvc_forward (one recurrence; the headline stays encode() + decode())",
           "value": round(B * (L / FS) * steps / dt, 2), "unit": "audio-seconds/s", "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps}
    if with_parity:                # the oracle's forward() = decode(encode(x)) on two utterances
        oc = get_oracle(conf)
        ref = oc.forward(x[:2].cpu(), bitrate)
        out["parity"] = {"waveform_rms_error_vs_oracle_forward": float((wav[:2].cpu() - ref).pow(2).mean().sqrt()),
                         "waveform_rms": float(ref.pow(2).mean().sqrt()), "utterances_checked": [0, 1]}
    return out


def leg_large_batch(conf, model, device, bitrate, with_parity, B=256, steps=3):
    """Not a BASELINE config: 256 x 5 s in ONE call, the persistent recurrence on interleaved chains (four utterance groups per
    workgroup) - what a caller gets who can batch more than 64 utterances."""
    L = int(FS * SECONDS)
    x = synth.synthetic_speech(B, L, seed=2000, kind="noise").to(device)

    def step():
        codes = model.encode(x, bitrate)
        return codes, model.decode(codes, L)
    step()
    dt, (codes, wav) = time_steps(step, steps, device)
    model.check_status()
    out = {"workload": f"batch {B} x {SECONDS:g} s in one call @ {bitrate:g} bit/s, full encode -> BigVGAN decode",
           "value": round(B * SECONDS * steps / dt, 2), "unit": "audio-seconds/s", "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps}
    if with_parity:                # the last utterance: last chain of the last workgroup slot
        out["parity"] = dict(spot_check(conf, model, x[B - 1:], codes[B - 1:], wav[B - 1:], L, bitrate), utterances_checked=[B - 1])
    del x, codes, wav
    torch.cuda.empty_cache()
    return out


def leg_encode_only(conf, model, device, x, bitrate, ref_codes, steps=5):
    """BASELINE configs[2]: STFT/mel + BVRNN.encode, no vocoder; `recurrence`: the persistent encode launch alone."""
    B, L = x.shape
    lib = _abi.load()
    model.encode(x, bitrate)
    dt, codes = time_steps(lambda: model.encode(x, bitrate), steps, device)
    _abi.check(lib.bvc_probe_begin(1, 1, 64))
    model.encode(x, bitrate)
    mean, mn, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int32()
    _abi.check(lib.bvc_probe_end(ctypes.byref(mean), ctypes.byref(mn), ctypes.byref(n)))
    model.check_status()
    fl = flops_per_step(conf, B, int(codes.shape[1]), "encode")[1][0]
    out = {"workload": f"BASELINE configs[2]: batch {B} x {L / FS:g} s, encode only (STFT/mel + BVRNN.encode), one batch at a time",
           "value": round(B * (L / FS) * steps / dt, 2), "unit": "audio-seconds/s", "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps,
           "recurrence": {"kernel": "bvrnn_flow_kernel<8,encode>", "launch_us": round(mean.value, 1), "gflop": round(fl / 1e9, 1),
                          "tflops": round(fl / (mean.value * 1e-6) / 1e12, 2) if mean.value else None,
                          "frac": round(fl / (mean.value * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if mean.value else None,
                          "timer": "hipEvent pair on the launch stream"}}
    if ref_codes is not None:      # the same input as the headline leg, whose codes the oracle checks: must be the same bits
        out["parity"] = {"codes_identical_to_headline_leg": bool(torch.equal(codes, ref_codes)),
                         "note": "the headline leg's `parity` block is the oracle check of these codes"}
    return out


def leg_streaming(conf, model, device, bitrate, with_parity, streams=256, hop=441, ticks=300):
    """BASELINE configs[4]: `streams` concurrent streams, 20 ms hops, one library call per hop (front-end -> encode -> decode ->
    incremental vocoder; the recurrences on the persistent kernel); host-observed latency per hop (synchronised)."""
    from bvcodec.streaming import StreamingCodec
    x = synth.synthetic_speech(streams, hop * ticks, seed=3, kind="noise").to(device)
    sc = StreamingCodec(model, streams, bitrate, hop=hop)
    lat, frames = [], 0
    keep_c, keep_w = [], []
    for i in range(ticks):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        c, w = sc.push(x[:, i * hop:(i + 1) * hop])
        torch.cuda.synchronize(device)
        lat.append(time.perf_counter() - t0)
        frames += c.shape[1]
        if with_parity and c.shape[1]:
            keep_c.append(c[:2].clone())
            keep_w.append(w[:2].clone())
    model.check_status()
    import numpy as np
    l = np.array(lat[50:]) * 1e3
    out = {"workload": f"BASELINE configs[4]: {streams} streams x {hop}-sample (20 ms) hops @ {bitrate:g} bit/s, encode + decode per hop, "
                       "one bvc_stream_codec_tick per hop (persistent recurrence kernel; BVC_STREAM_FLOW=0: a replayed hipGraph of launch-per-layer kernels)",
           "p50_ms": round(float(np.percentile(l, 50)), 3), "p99_ms": round(float(np.percentile(l, 99)), 3), "mean_ms": round(float(l.mean()), 3),
           "hop_budget_ms": round(1e3 * hop / FS, 2), "ticks": ticks, "ticks_timed": int(l.size), "frames_per_hop": round(frames / ticks, 3),
           "value": round(streams * (hop / FS) / (float(l.mean()) * 1e-3), 1), "unit": "audio-seconds/s (all streams, at the mean hop latency)"}
    if with_parity:
        codes, wav = torch.cat(keep_c, 1), torch.cat(keep_w, 1)
        F = codes.shape[1]
        out["parity"] = dict(spot_check(conf, model, x[:2], codes, wav, 256 * F, bitrate, frames=F), streams_checked=[0, 1], frames=F)
    del sc
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help="utterances per GPU")
    ap.add_argument("--seconds", type=float, default=SECONDS)
    ap.add_argument("--bitrate", type=float, default=BITRATE)
    ap.add_argument("--mode", choices=["codec", "encode"], default="codec",
                    help="codec: encode -> BigVGAN decode (configs[1]); encode: STFT/mel + BVRNN.encode only (configs[2])")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams of the HEADLINE measurement (1: one batch at a time; the 4-stream figure is reported "
                         "beside it as multi_stream)")
    ap.add_argument("--multi-streams", type=int, default=4, help="streams of the multi_stream figure (0: skip)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the target_workload / encode_only / streaming / large_batch legs")
    ap.add_argument("--legs", default="target,encode,fused,streaming,large", help="which of the extra legs run (comma separated)")
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
    full = a.mode == "codec"
    x = synth.synthetic_speech(B, L, seed=rank, kind="noise").to(device)      # resident before timing
    use_pg = torch.distributed.is_available() and torch.distributed.is_initialized()
    nslots = max(1, a.streams, a.multi_streams)
    gathered = ([torch.empty(world * B, L, device=device) for _ in range(nslots)]
                if (use_pg and full and not a.no_gather) else None)

    def local_step():
        codes = model.encode(x, a.bitrate)
        return codes, (model.decode(codes, L) if full else None)

    gather_ev = []                                 # (start, end) HIP events around every gather, on the stream it was issued on

    def step(slot=0):
        codes, wav = local_step()
        if gathered is not None:                  # the "final gather" of the north star: one RCCL collective
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            torch.distributed.all_gather_into_tensor(gathered[slot], wav)
            e1.record()
            gather_ev.append((e0, e1))
            bdist.fence_collective(device)        # a persistent launch of ANY stream waits for the collective (it holds CUs while it waits for peers)
        return codes, wav

    def run(n, streams, fn=step):
        """n steps round-robin on `streams` (one stream: back to back); every step completes before run() returns"""
        last = None
        for k in range(n):
            with torch.cuda.stream(streams[k % len(streams)]):
                last = fn(k % len(streams))
        for st in streams:
            torch.cuda.current_stream(device).wait_stream(st)
        return last

    def pick_streams(n):
        # streams that really are concurrent: which HIP streams share a hardware queue is not visible through the API and
        # shifts when an RCCL communicator exists, so the set is measured (bvcodec.dist.concurrent_stream_sets, ~0.3 s) and
        # the candidate sets are settled with the real workload (local steps only: no collective in the probe)
        if n <= 1:
            return [torch.cuda.current_stream(device)]
        try:
            sets = bdist.concurrent_stream_sets(n, device)
        except Exception as e:                      # never let the placement probe take the benchmark down
            print(f"warning: stream placement probe failed ({e!r}); using the first {n} streams", file=sys.stderr)
            return [torch.cuda.Stream(device) for _ in range(n)]
        best = None
        for cand in sets:
            run(len(cand), cand, lambda _s: local_step())
            torch.cuda.synchronize(device)
            t_try = time.perf_counter()
            run(2 * len(cand), cand, lambda _s: local_step())
            torch.cuda.synchronize(device)
            t_try = time.perf_counter() - t_try
            if best is None or t_try < best[0]:
                best = (t_try, cand)
        return best[1]

    def timed(n, streams, fn=step):
        if use_pg:
            torch.distributed.barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        last = run(n, streams, fn)
        torch.cuda.synchronize(device)
        if use_pg:
            torch.distributed.barrier()
        return time.perf_counter() - t0, last

    streams = pick_streams(a.streams)
    if a.warmup:
        run(max(a.warmup, len(streams)), streams)
    gather_ev.clear()
    elapsed_local, (codes, wav) = timed(a.steps, streams)
    gather_ms = (sum(e0.elapsed_time(e1) for e0, e1 in gather_ev) / len(gather_ev)) if gather_ev else 0.0
    elapsed, rank_ms = elapsed_local, [round(1e3 * elapsed_local / a.steps, 3)]
    if use_pg:
        tl = torch.tensor([elapsed_local], device=device, dtype=torch.float64)
        tall = [torch.zeros_like(tl) for _ in range(world)]
        torch.distributed.all_gather(tall, tl)
        rank_ms = [round(1e3 * float(t.item()) / a.steps, 3) for t in tall]
        elapsed = max(float(t.item()) for t in tall)             # the job is as slow as its slowest rank
    assert codes.shape[2] == conf["z_dim"] and (wav is None or torch.isfinite(wav).all())
    model.check_status()                                          # no recurrence kernel may have given up waiting

    T = codes.shape[1]
    names = {"codec": "full encode -> BigVGAN decode", "encode": "encode only (STFT/mel + BVRNN.encode, no vocoder)"}
    cfgname = ("custom" if B != BATCH else "configs[1]" if (full and a.seconds == 5.0) else "configs[2]" if a.seconds == 5.0 else
               "configs[3] shard (one GPU's 64 of 512 x 10 s)" if (full and a.seconds == 10.0) else "custom")
    out = {
        "metric": "audio-seconds coded per wall-second (encode+decode), 22.05 kHz @ 3 kbit/s" if (full and a.bitrate == BITRATE) else
                  f"audio-seconds coded per wall-second ({'encode+decode' if full else 'encode only'}), 22.05 kHz @ {a.bitrate:g} bit/s",
        "value": round(world * B * a.seconds * a.steps / elapsed, 2),
        "unit": "audio-seconds/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1e3 * elapsed / a.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE {cfgname}: batch {B} x {a.seconds:g} s utterances per GPU, config_varBitRate @ "
                               f"{a.bitrate:g} bit/s ({int(model.bits_per_frame(a.bitrate))} bits/frame), {names[a.mode]}; "
                               f"{'one batch at a time on one stream' if len(streams) == 1 else str(len(streams)) + ' batches in flight on ' + str(len(streams)) + ' streams'}",
                   "frames_per_utterance": T, "weights": "seeded synthetic (checkpoints are LFS pointers)",
                   "gather": "rccl all_gather of decoded waveforms" if gathered is not None else "none",
                   "streams": len(streams),
                   "library_options": {k: model.engine().get_option(k) for k in ("recurrence", "encode_fold", "decode_fold")}},
        "rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms), "per_rank": rank_ms},
        "gather_ms": round(gather_ms, 3),          # rank 0's mean all-gather time per step (0: no process group, nothing to gather)
    }

    if a.multi_streams > 1 and len(streams) == 1:
        # the same K steps with several independent batches in flight (throughput-oriented; results are bit-identical to the
        # serial schedule: tests/test_gpu_concurrency.py).  Timed like the headline (barrier + synchronize, max over ranks).
        def multi():
            ms = pick_streams(a.multi_streams)             # (settled with the schedule that is about to be timed)
            run(len(ms), ms)
            dt, _ = timed(a.steps, ms)
            if use_pg:
                tm = torch.tensor([dt], device=device, dtype=torch.float64)
                torch.distributed.all_reduce(tm, op=torch.distributed.ReduceOp.MAX)
                dt = float(tm.item())
            return {"value": round(world * B * a.seconds * a.steps / dt, 2), "ms_per_step": round(1e3 * dt / a.steps, 3)}

        res = multi()        # no schedule switch here: the library's default ("auto") sees the overlapping calls and takes the layer kernels
        model.check_status()
        out["multi_stream"] = dict(res, streams=a.multi_streams,
                                   note=f"{a.multi_streams} x {B} utterances in flight per GPU, same steps round-robin on {a.multi_streams} HIP streams; "
                                        "recurrence schedule left at the library default (auto: launch per layer while calls of several streams overlap)")
        for _ in range(3):   # back to one batch at a time: the default returns to the persistent kernel after two calls without company
            local_step()
        torch.cuda.synchronize(device)

    if rank == 0 and not a.no_roofline:
        lib = _abi.load()
        fam = flops_per_step(conf, B, T, a.mode, fold=bool(model.engine().get_option("decode_fold")))
        rows = {}
        for kind in ((1, 3, 4, 5, 6) if full else (1, 4, 5)):               # HIP event pairs around every launch of the family
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
        # the committed rocprofv3 summary is of the DEFAULT command: only that workload is comparable with it
        default_cmd = full and a.seconds == SECONDS and a.bitrate == BITRATE and B == BATCH
        rp = rocprof_avg_us("bvrnn_flow_kernel") if (dom == 1 and default_cmd) else None
        fl, launches = fam[dom]
        out["roofline"] = {"bound": "mfma", "achieved": r["tflops"], "peak": PEAK_FP32_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": round(r["tflops"] / PEAK_FP32_MFMA_TFLOPS, 4),
                           "traffic": pmc_traffic_bytes("bvrnn_flow_kernel") if dom == 1 else None,
                           "traffic_unit": "HBM-side bytes per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate pass: "
                                           f"profiles/{os.path.basename(PMC_SUMMARY)}); algorithmic: the weights once per frame "
                                           "that are not resident on the chip (80 / 53 MB encode / decode, served by the 256 MB Infinity Cache) + 0.25 MB per audio-second of I/O",
                           "kernel": r["kernel"], "mean_launch_us": r["mean_us"],
                           "launches_per_step": r["launches_per_step"], "gflop_per_launch": round(fl / launches / 1e9, 1),
                           "timer": r["timer"],
                           "rocprof_avg_us": rp,
                           "frac_rocprof": round(fl / launches / (rp * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if rp else None,
                           "rocprof_source": f"profiles/{os.path.basename(ROCPROF_SUMMARY)} (rocprofv3 --kernel-trace --stats of this command)",
                           "note": "fp32-in/fp32-acc MFMA (v_mfma_f32_16x16x4_f32).  One launch runs every layer of every frame; it leans on "
                                   "the hand-offs between dependent layers (a flag poll and an operand fetch through the fabric per layer) and on "
                                   "the L2 -> CU fill path (l2_to_cu), not on the matrix pipe (busy 39 % of the time) or on HBM (DESIGN.md 4)"}
        l2b = pmc_l2_to_cu_bytes("bvrnn_flow_kernel") if dom == 1 else None
        if l2b:
            out["roofline"]["l2_to_cu"] = {"bytes_per_launch": l2b, "tb_per_s": round(l2b / (r["mean_us"] * 1e-6) / 1e12, 2),
                                           "source": f"TCP_TCC_READ_REQ_sum x 128 B, profiles/{os.path.basename(PMC_SUMMARY)} (separate --pmc pass), over this run's launch time"}
        # SURVEY.md 8(d) also asks for the whole path against the fp32 peak: algorithmic FLOPs of a step (all families)
        # over the measured step time of the timed region
        step_flops = sum(f for f, _ in fam.values())
        whole = step_flops / (elapsed / a.steps) / 1e12
        out["roofline"]["whole_path"] = {"achieved": round(whole, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                         "frac": round(whole / PEAK_FP32_MFMA_TFLOPS, 4),
                                         "gflop_per_audio_second": round(step_flops / (B * a.seconds) / 1e9, 3)}
        out["kernel_families"] = rows
    default_workload = full and a.seconds == SECONDS and B == BATCH
    if rank == 0:
        note("headline, multi_stream and roofline legs done")
    legs = set() if a.no_extra else set(a.legs.split(","))
    if default_workload and "target" in legs and (use_pg or rank == 0):
        # with a process group every rank runs its shard of configs[3] (the gather is a collective); rank 0 reports
        tw = leg_target_workload(conf, model, device, B, a.bitrate, not a.no_parity, world=world, use_pg=use_pg and not a.no_gather, rank=rank)
        if rank == 0:
            out["target_workload"] = tw
            note("target_workload done")
    if rank == 0 and world == 1 and default_workload:
        with_par = not a.no_parity
        if "encode" in legs:
            out["encode_only"] = leg_encode_only(conf, model, device, x, a.bitrate, codes)
            note("encode_only done")
        if "fused" in legs:
            out["forward_fused"] = leg_forward_fused(conf, model, device, x, a.bitrate, with_par)
            note("forward_fused done")
        if "streaming" in legs:
            out["streaming"] = leg_streaming(conf, model, device, a.bitrate, with_par)
            note("streaming done")
        if "large" in legs:
            out["large_batch"] = leg_large_batch(conf, model, device, a.bitrate, with_par)
            note("large_batch done")
    if rank == 0 and not (a.no_parity and (a.no_cpu_baseline or world > 1)):
        leg = oracle_leg(conf, model, x, codes, wav, L, a.bitrate, a.mode, with_baseline=(world == 1 and not a.no_cpu_baseline))
        if not a.no_parity:
            out["parity"] = leg["parity"]
        if "cpu_baseline" in leg:
            out["cpu_baseline"] = leg["cpu_baseline"]
        note("parity / cpu_baseline done")
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_pg:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
