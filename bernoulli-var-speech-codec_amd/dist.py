"""Multi-GPU sharding of a batch of independent utterances (one process per GPU).

The reference is single-device; utterances are independent (no cross-batch op at
bvrnn.py:186-206,222-227 or models.py:207-238), so a batch shards trivially: each rank encodes and
decodes its contiguous slice with zero communication and one RCCL collective per gathered tensor (waveforms;
optionally the codes too) collects the results
(``torch.distributed`` backend "nccl" is RCCL on ROCm; over xGMI).  On CPU-only hosts the same code
runs on gloo, which is how tests/test_dist_cpu.py covers it.
"""
import os

import torch
import torch.distributed as dist


def init_from_env():
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns
    (rank, world_size, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
    force = os.environ.get("BVC_FORCE_PG", "0") == "1"      # lets a 1-GPU box exercise the RCCL code path
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if use_cuda:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    return rank, world, device


def fence_collective(device=None):
    """Call right behind a collective issued on the current stream: the next persistent recurrence launch (which needs every
    compute unit) then waits for it - an RCCL kernel that is still waiting for its peers holds compute units.  A no-op
    without a GPU; see bvc_flow_fence in include/bvcodec.h."""
    if not torch.cuda.is_available():
        return
    from . import _abi
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(dev):
        _abi.check(_abi.load().bvc_flow_fence(_abi.current_stream(dev)))


def shard_range(total, world, rank):
    """Contiguous split of `total` utterances; the first total % world ranks take one more."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_batch(local, total=None, out=None):
    """All-gather the per-rank slices (dim 0) of a batch into the full batch on every rank.
    Equal slices use one all_gather_into_tensor; ragged slices are padded to the largest."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    if total is None or total % world == 0:
        if out is None:
            out = local.new_empty((local.shape[0] * world,) + tuple(local.shape[1:]))
        dist.all_gather_into_tensor(out, local.contiguous())
        if local.is_cuda:
            fence_collective(local.device)
        return out
    sizes = [shard_range(total, world, r) for r in range(world)]
    nmax = max(hi - lo for lo, hi in sizes)
    pad = local.new_zeros((nmax,) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    buf = local.new_empty((nmax * world,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(buf, pad)
    if local.is_cuda:
        fence_collective(local.device)
    return torch.cat([buf[r * nmax: r * nmax + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], 0)


def codec_sharded(model, x, bitrate, gather=True, gather_codes=True):
    """encode+decode of the rank's slice of x (B_total, L) and (optionally) the gathered result.
    Returns (codes, wav) - full batch on every rank when gather=True, local slice otherwise.
    gather=True issues one collective for the waveforms and, unless gather_codes=False, a second one for the
    codes (then the returned codes stay the local slice).  Every rank needs at least one utterance: with fewer
    utterances than ranks ALL ranks raise before any work or collective is issued (an empty rank would
    otherwise fail alone and leave the others waiting in the gather)."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    if x.shape[0] < world:
        raise ValueError(f"codec_sharded: {x.shape[0]} utterances cannot be sharded over {world} ranks "
                         "(every rank needs at least one)")
    lo, hi = shard_range(x.shape[0], world, rank)
    xl = x[lo:hi]
    codes = model.encode(xl, bitrate)
    wav = model.decode(codes, x.shape[1])
    if gather and world > 1:
        return (gather_batch(codes, x.shape[0]) if gather_codes else codes), gather_batch(wav, x.shape[0])
    return codes, wav


def concurrent_stream_sets(n, device, candidates=8, chain=200, max_sets=3):
    """Candidate sets of n torch streams on `device` that really run concurrently, best guess first.

    HIP multiplexes its streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default); which streams share one is
    not visible through the API and changes with what was created before (an RCCL communicator shifts it).  Two streams on
    one queue serialise, which costs the multi-stream schedule of bench.py a quarter of its throughput.  So: measure.
    Every pair of `candidates` fresh streams runs two dependent chains of tiny kernels; a pair that takes about as long as
    two chains back to back shares a queue.  Returns up to `max_sets` sets of n mutually concurrent candidates (sets of
    streams with fewer conflicts first; the first n candidates if nothing conclusive is found).  Costs ~0.3 s once.
    A caller with a real workload should time the sets and keep the best (bench.py does)."""
    import itertools
    import time
    if device.type != "cuda":
        return [[]]
    if n <= 1:
        return [[torch.cuda.Stream(device)]]
    cands = [torch.cuda.Stream(device) for _ in range(max(candidates, n))]
    bufs = [torch.zeros(256, device=device) for _ in cands]

    def run(idx):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(chain):
            for i in idx:
                with torch.cuda.stream(cands[i]):
                    bufs[i].add_(1.0)
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0

    run([0, 1])                                                   # warm-up (stream creation, kernel load)
    # issuing 2 x chain launches from one host thread is itself the floor for a concurrent pair: compare pairs with each
    # other, not with a single chain
    k = len(cands)
    t = {}
    for i in range(k):
        for j in range(i + 1, k):
            t[(i, j)] = min(run([i, j]) for _ in range(2))
    fastest = min(t.values())
    conflict = {p for p, v in t.items() if v > 1.5 * fastest}
    score = {i: sum(1 for p in conflict if i in p) for i in range(k)}
    sets = [c for c in itertools.combinations(range(k), n)
            if not any(p in conflict for p in itertools.combinations(c, 2))]
    sets.sort(key=lambda c: (sum(score[i] for i in c), c))
    if not sets:
        sets = [tuple(range(n))]
    # prefer sets that differ from each other (so that timing them tells something)
    out = [sets[0]]
    for c in sets[1:]:
        if len(out) >= max_sets:
            break
        if all(len(set(c) & set(o)) < n - 1 for o in out):
            out.append(c)
    return [[cands[i] for i in c] for c in out]


def concurrent_streams(n, device, candidates=8, chain=200):
    """The first candidate set of concurrent_stream_sets()."""
    return concurrent_stream_sets(n, device, candidates, chain, max_sets=1)[0]
