"""world_size-2 gloo run of the sharding / gather helpers (bvcodec/dist.py) on CPU."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _FakeCodec:
    """Stand-in with the facade's encode/decode signature (the real one needs the GPU)."""

    def encode(self, x, bitrate):
        return (x[:, :8, None] * torch.arange(4)[None, None, :]).contiguous()

    def decode(self, codes, length):
        return codes.sum(-1).repeat(1, length // 8 + 1)[:, :length].contiguous()


def _worker(rank, world, port, total, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from bvcodec import dist as bdist
    r, w, dev = bdist.init_from_env()
    assert (r, w) == (rank, world) and dev.type == "cpu"
    x = torch.arange(total * 16, dtype=torch.float32).reshape(total, 16)
    codes, wav = bdist.codec_sharded(_FakeCodec(), x, 3000, gather=True)
    ref_c = _FakeCodec().encode(x, 3000)
    ref_w = _FakeCodec().decode(ref_c, 16)
    ok = torch.equal(codes, ref_c) and torch.equal(wav, ref_w)
    lo, hi = bdist.shard_range(total, world, rank)
    local_c, _ = bdist.codec_sharded(_FakeCodec(), x, 3000, gather=False)
    ok = ok and torch.equal(local_c, ref_c[lo:hi])
    out_q.put((rank, bool(ok), lo, hi))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_even_shards_gather():
    res = _run(8)
    assert res == [(0, True, 0, 4), (1, True, 4, 8)]


def test_ragged_shards_gather():
    res = _run(7)
    assert res == [(0, True, 0, 4), (1, True, 4, 7)]


def test_shard_range_covers_batch():
    sys.path.insert(0, ROOT)
    from bvcodec import dist as bdist
    for total in (1, 7, 64, 512):
        for world in (1, 2, 4, 8):
            spans = [bdist.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_concurrent_stream_sets_without_gpu():
    """The stream picker is a no-op off the GPU (bench.py only calls it on the MI355X)."""
    import torch
    from bvcodec import dist as bdist
    assert bdist.concurrent_stream_sets(3, torch.device("cpu")) == [[]]


def test_more_ranks_than_utterances_raises_everywhere():
    """Every rank must refuse BEFORE any compute or collective (otherwise the empty rank fails alone and the others hang
    in the gather).  Without a process group world = 1, so emulate the check through the public helper."""
    import pytest
    sys.path.insert(0, ROOT)
    from bvcodec import dist as bdist
    x = torch.zeros(0, 16)
    with pytest.raises(ValueError):
        bdist.codec_sharded(_FakeCodec(), x, 3000)


def test_default_checkpoint_paths_are_reported(tmp_path, monkeypatch):
    """BVRNNCodecModel() with the reference's default arguments: the Git-LFS checkpoints are not shipped, so the
    constructor must say which file it looked for and where to put it."""
    import pytest
    sys.path.insert(0, ROOT)
    from bvcodec import model as bmodel
    assert os.path.basename(bmodel.default_chkpt_bvrnn) == "bvrnn_var_bitrate_step200000"
    assert os.path.basename(bmodel.default_chkpt_vocoder) == "bigvgan_causal_tiny_ftbvrnn_g_step3500000"
    if not os.path.exists(bmodel.default_chkpt_bvrnn):
        with pytest.raises(FileNotFoundError, match="BVC_CHKPT_DIR"):
            bmodel.BVRNNCodecModel()
