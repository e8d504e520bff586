"""CPU, world_size 2 over gloo: the N > 1 path of bench.py (read sharding + final gather of GAF text)."""
import os
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from recgraph_amd.shard import gather_text, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    reads = ["read%04d" % i for i in range(11)]
    lo, hi = shard_bounds(len(reads), rank, world)
    text = "".join(r + "\tGAF\n" for r in reads[lo:hi]).encode()
    parts = gather_text(text, rank, world, device="cpu")
    if rank == 0:
        q.put(b"".join(parts).decode())
    else:
        assert parts is None
    empty = gather_text(b"" if rank == 1 else b"x", rank, world, device="cpu")
    if rank == 0:
        q.put(empty)
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    from recgraph_amd.shard import shard_bounds
    for n in (0, 1, 7, 8, 100001):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gather_preserves_read_order():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    text = q.get(timeout=120)
    empty = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert text == "".join("read%04d\tGAF\n" % i for i in range(11))
    assert empty == [b"x", b""]
