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


def _run_bench(extra, env_extra):
    import json
    import subprocess
    env = dict(os.environ, RG_BENCH_STUB="1", **env_extra)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    base = ["--config", "C2", "--batch", "64", "--steps", "3", "--warmup", "1", "--no-cpu"]
    for k in range(0, len(extra), 2):            # a flag given in `extra` replaces the default one
        if extra[k] in base:
            i = base.index(extra[k])
            del base[i:i + 2]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + base + extra, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout          # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_launcher_runs_the_world_2_path():
    """`python bench.py --gpus 2` with no launcher in the environment starts two rank processes itself and runs the
    real world > 1 code path of bench.py (process group, per-rank read shards, barrier + max-over-ranks timing, gather
    of every step's text to rank 0) — over gloo with the device work stubbed out (RG_BENCH_STUB=1: there is no GPU
    here), same code otherwise."""
    weak = _run_bench(["--gpus", "2"], {})
    assert weak["n_gpus"] == 2 and weak["scaling"] == "weak" and weak["steps"] == 3 and "STUB" in weak["data"]
    assert "128 reads/step over all GPUs" in weak["config"]["workload"]
    # the stub formats 'read<i>\t<16 bases>\n' per read: all 3 steps of both ranks must arrive on rank 0
    per_rank = 3 * sum(len("read%d\t" % i) + 16 + 1 for i in range(64))
    assert weak["gaf_bytes_gathered"] == 2 * per_rank
    assert "64-read tiles" in weak["config"]["workload"]
    # the stated target beside the headline, in the same run (the driver passes no flags): the 25 x 64 canonical reads split
    # EVENLY over the ranks (reads, not tiles), even tiles of at most 64 reads
    s100 = weak["strong_100k"]
    # (13 even tiles, the first cut in two — a short one first — so that the handles start out of phase: --strong-ramp 1)
    assert s100["reads"] == 25 * 64 and s100["reads_per_rank"] == [800, 800] and s100["tiles_per_rank"] == [14, 14]
    assert sum(s100["tile_reads"]) == 800 and max(s100["tile_reads"]) <= 64 and s100["tile_reads"][0] < s100["tile_reads"][1] and s100["reads_per_s"] > 0
    assert max(s100["tile_reads"][2:]) - min(s100["tile_reads"][2:]) <= 1
    # --scaling strong: the steps x batch reads of the N = 1 run split evenly: 192 reads -> 96 per rank -> two tiles of 48
    tile = lambda k: sum(len("read%d\t" % i) + 16 + 1 for i in range(k))
    strong = _run_bench(["--gpus", "2", "--scaling", "strong"], {})
    assert strong["n_gpus"] == 2 and strong["scaling"] == "strong" and strong["strong_100k"] is None
    assert "64 reads/step over all GPUs" in strong["config"]["workload"] and "48-read tiles" in strong["config"]["workload"]
    assert strong["tiles_per_rank"] == [2, 2] and strong["gaf_bytes_gathered"] == 4 * tile(48)
    assert "gather_ms_per_step" in strong and "gather_wait_ms" in strong and strong["host_cpu_s_per_step"] >= 0
    one = _run_bench([], {})
    assert one["n_gpus"] == 1 and one["gaf_bytes_gathered"] == per_rank and one["strong_100k"] is None
    # an odd read count: the shares differ by one read, never by a tile
    odd = _run_bench(["--gpus", "2", "--scaling", "strong", "--steps", "1", "--batch", "65"], {})
    assert odd["tiles_per_rank"] == [1, 1] and odd["gaf_bytes_gathered"] == tile(33) + tile(32)


def test_bench_launcher_runs_eight_ranks():
    """The driver's N = 8 shape on the CPU (device work stubbed): eight rank processes, one process group, even read shards,
    the 8-way gather of every step's text to rank 0 (the modelled host budget is in profiles/r05_world8_stub.json)."""
    d = _run_bench(["--gpus", "8", "--batch", "16", "--steps", "2"], {})
    assert d["n_gpus"] == 8 and d["steps"] == 2 and "STUB" in d["data"]
    per_rank = 2 * sum(len("read%d\t" % i) + 16 + 1 for i in range(16))
    assert d["gaf_bytes_gathered"] == 8 * per_rank
    s100 = d["strong_100k"]
    assert s100["reads"] == 25 * 16 and s100["reads_per_rank"] == [50] * 8 and len(s100["tiles_per_rank"]) == 8


def test_even_tiles():
    from recgraph_amd.shard import even_tiles, shard_bounds
    assert even_tiles(12800, 4096) == [3200] * 4 and even_tiles(4096, 4096) == [4096] and even_tiles(0, 4096) == []
    assert even_tiles(4097, 4096) == [2048, 2049]
    for n in (1, 5, 4095, 12800, 102400, 99999):
        for ramp in (0, 1, 3):
            t = even_tiles(n, 4096, ramp)
            assert sum(t) == n and max(t) <= 4096 and min(t) >= 1
    # 102 400 reads on 8 ranks: 12 800 each, four launches each (whole 4096-read tiles would give 4/3/3/3/3/3/3/3)
    spans = [shard_bounds(102400, r, 8) for r in range(8)]
    assert {b - a for a, b in spans} == {12800} and {len(even_tiles(b - a, 4096)) for a, b in spans} == {4}
    r = even_tiles(12800, 4096, ramp=2)
    assert len(r) == 6 and r[0] < r[1] and sum(r) == 12800


def test_bench_refuses_a_world_size_it_was_not_asked_for():
    import subprocess
    env = dict(os.environ, RG_BENCH_STUB="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-cpu"], env=env, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 2 and "WORLD_SIZE" in out.stderr


def test_bench_step_pipeline_shapes():
    """One handle (steps back to back) or several: every timed step's text arrives exactly once, in order."""
    per_rank = 3 * sum(len("read%d\t" % i) + 16 + 1 for i in range(64))
    for extra in (["--handles", "1"], ["--handles", "2"], ["--handles", "3"]):
        d = _run_bench(extra, {})
        assert d["n_gpus"] == 1 and d["gaf_bytes_gathered"] == per_rank, extra


def _fixed_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from recgraph_amd.shard import FixedGather
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = FixedGather(rank, world, "cpu", cap=64, depth=3)
    steps = 8                                  # more steps than ring slots: every slot is reused
    for s in range(steps):
        data = (b"" if (s + rank) % 3 == 0 else ("r%d-s%d;" % (rank, s)).encode() * (1 + (s + rank) % 5))
        g.submit(data)
    res, sizes = g.finish()
    if rank == 0:
        q.put([[bytes(p.numpy().tobytes()) for p in step] for step in res])
        q.put(sizes)
    else:
        assert res is None
    # a payload beyond the agreed capacity is refused on the rank that holds it, before any collective is issued
    try:
        FixedGather(rank, world, "cpu", cap=8).submit(b"x" * 9)
        q.put("no error")
    except ValueError:
        q.put("refused")
    dist.destroy_process_group()


def test_fixed_capacity_gather_over_gloo():
    """`FixedGather` (the per-step gather without host synchronisation on ranks > 0): one collective per step,
    lengths in a header, a ring of staging buffers that is reused — every byte arrives, in step order, empty payloads too."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_fixed_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(4)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    res = next(x for x in got if isinstance(x, list) and x and isinstance(x[0], list) and x[0] and isinstance(x[0][0], bytes))
    sizes = next(x for x in got if isinstance(x, list) and x and isinstance(x[0], list) and isinstance(x[0][0], int))
    exp = [[(b"" if (s + r) % 3 == 0 else ("r%d-s%d;" % (r, s)).encode() * (1 + (s + r) % 5)) for r in range(2)] for s in range(8)]
    assert res == exp
    assert sizes == [[len(x) for x in step] for step in exp]
    assert got.count("refused") == 2
