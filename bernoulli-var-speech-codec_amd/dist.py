"""Multi-GPU sharding of a batch of independent utterances (one process per GPU).

The reference is single-device; utterances are independent (no cross-batch op at
bvrnn.py:186-206,222-227 or models.py:207-238), so a batch shards trivially: each rank encodes and
decodes its contiguous slice with zero communication and ONE RCCL collective gathers the results
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
        return out
    sizes = [shard_range(total, world, r) for r in range(world)]
    nmax = max(hi - lo for lo, hi in sizes)
    pad = local.new_zeros((nmax,) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    buf = local.new_empty((nmax * world,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(buf, pad)
    return torch.cat([buf[r * nmax: r * nmax + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], 0)


def codec_sharded(model, x, bitrate, gather=True):
    """encode+decode of the rank's slice of x (B_total, L) and (optionally) the gathered result.
    Returns (codes, wav) - full batch on every rank when gather=True, local slice otherwise."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_range(x.shape[0], world, rank)
    xl = x[lo:hi]
    codes = model.encode(xl, bitrate)
    wav = model.decode(codes, x.shape[1])
    if gather and world > 1:
        return gather_batch(codes, x.shape[0]), gather_batch(wav, x.shape[0])
    return codes, wav
