#!/usr/bin/env python3
"""bench.py — headline benchmark of the RecGraph DP hot path on MI355X.

Workload (BASELINE.json configs[4], the configuration the north-star target is quoted on): -m 8 recombination
alignment (R=4 r=0.1 B=1) of synthetic 1 kbp reads against a fixed synthetic ~10 k-row / 32-path graph.  The read set is
25 distinct batches of 4096 reads (102 400 distinct reads per GPU, seeded); a "step" is one pass of the hot path over ONE
batch: upload of the batch's reads (rg_batch_set_reads: the timed region starts with the graph resident and the reads in
host memory in the C ABI's input form, bases + offsets — 4 MB per step, PCIe-inclusive), two DP sweeps, candidate expansion, search,
layer rebuild and traceback on the device, record fetch and GAF formatting on the host.  Consecutive steps rotate over
three batch handles (work-buffer sets in HBM, `--handles`) whose device parts run on three host threads / streams
(`--device-threads`): the latency-bound small kernels of one step fill the gaps of the other steps' sweeps, and the host
formatting and the upload of later steps overlap all of it.  The GAF text of EVERY timed step is kept and gathered to
rank 0 at the end (inside the timed region), over RCCL when N > 1.

Multi-GPU: reads shard across ranks (one process per GPU, graph replicated, no data-path collective).  Launched by
torchrun (RANK / LOCAL_RANK / WORLD_SIZE in the environment) or directly: `python bench.py --gpus N` starts the N rank
processes itself before anything touches the GPU.  `--scaling weak` (default): every rank aligns steps x batch reads of
its own; `--scaling strong`: the steps x batch reads of the N=1 run are sharded over the ranks (shard_bounds).

After the timed region rank 0 compares the GAF text of reads of the last timed step byte for byte with the CPU
restatement's (the in-run parity gate: exit status 3 and "parity_ok": false on any difference) and times the CPU legs
of `cpu_baseline`.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_CELL_UPDATE = {0: 12, 2: 32, 4: 8, 8: 12}   # SURVEY §8d algorithmic bytes per unit of work
HBM_PEAK_GBS = 8000.0                                 # MI355X_MICROARCH.md: 8.0 TB/s spec
DISTINCT_BATCHES = 25                                 # 25 x 4096 = 102 400 distinct reads per GPU
DEFAULT_BATCH = {"C2": 10000, "C3": 10000, "C4": 4096, "C5": 4096}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=25, help="timed steps (default: 25 x 4096 = 102 400 distinct reads, BASELINE.json's 100 k)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C5", choices=["C2", "C3", "C4", "C5"])
    ap.add_argument("--batch", type=int, default=0, help="reads per step per GPU (default: per config)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--cpu-reads", type=int, default=-1, help="cap on the reads of every cpu_baseline leg (0 = skip the legs)")
    ap.add_argument("--no-cpu", action="store_true", help="skip cpu_baseline and the parity gate")
    ap.add_argument("--handles", type=int, default=3, help="batch handles (work-buffer sets in HBM) the steps rotate over")
    ap.add_argument("--device-threads", type=int, default=3, choices=[1, 2, 3, 4],
                    help="host threads (and streams) that run the device parts of the handles concurrently; 1 = one after the other")
    return ap.parse_args(argv)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N rank processes (children of a parent that never
    touches the GPU) and return rank 0's exit status.  Fails loudly when a rank fails."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rcs = [None] * len(procs)
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
        if any(rc not in (None, 0) for rc in rcs):
            for i, p in enumerate(procs):           # a rank failed: stop the others (they would wait in a collective)
                if rcs[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    rcs[i] = p.wait()
            break
        time.sleep(0.05)
    bad = [(i, rc) for i, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write("bench.py: ranks failed: %s\n" % bad)
        return bad[0][1] or 1
    return 0


class HipEngine:
    """The product: `nhandles` rg_batch handles on this rank's GPU behind the C ABI (ctypes)."""

    def __init__(self, dev, gfa, mode, first_reads, nhandles=2):
        from recgraph_amd import _lib, api
        self._lib, self._api = _lib, api
        _lib.check(_lib.load().rg_set_device(dev))
        self.dev = dev
        self.graph = api.Graph.from_gfa_text(gfa)
        self.params = api.make_params(mode)
        self.handles = [api.Batch(self.graph, first_reads, self.params)]
        try:
            # setup (not a timed or warm-up step): the other handles and the work buffers of all; handles whose buffers
            # do not fit the HBM are dropped (with one handle the steps run back to back)
            self.handles[0].run()
            for _ in range(1, nhandles):
                self.handles.append(api.Batch(self.graph, first_reads, self.params))
                self.handles[-1].run()
        except _lib.RecGraphError:
            del self.handles[-1]
        self.rows, self.paths = self.graph.rows, self.graph.paths_number
        self.host_s = {}

    def thread_init(self):
        self._lib.check(self._lib.load().rg_set_device(self.dev))     # hipSetDevice is per thread

    def pack(self, reads):
        return self._api.Batch.pack_reads(reads)        # the C ABI's input form: bases blob + offsets

    def set_reads(self, h, packed):
        t0 = time.perf_counter()
        h.set_reads(packed)     # canonicalise + upload, inside the step (main thread: overlaps the other handle's kernels)
        self.host_s["set_reads"] = self.host_s.get("set_reads", 0.0) + time.perf_counter() - t0

    def device_part(self, h):
        t1 = time.perf_counter()
        h.run()
        t2 = time.perf_counter()
        h.fetch()
        t3 = time.perf_counter()
        for k, v in (("run", t2 - t1), ("fetch", t3 - t2)):
            self.host_s[k] = self.host_s.get(k, 0.0) + v
        return h

    def format(self, h, nthreads):
        return h.format_all(None, 1, nthreads)

    def sync(self):
        import torch
        torch.cuda.synchronize()


class StubEngine:
    """RG_BENCH_STUB=1: NO device work — lets the CPU test suite run this file's launcher, sharding, barrier/all-reduce
    timing and text gather (the world > 1 code path) over gloo.  Its JSON line says so; it is not a measurement."""

    class H:
        cell_updates = 0

        def kernel_stats(self):
            return {}

    def __init__(self, dev, gfa, mode, first_reads, nhandles=2):
        self.host_s = {}
        self.handles = [self.H() for _ in range(nhandles)]
        self.rows, self.paths = gfa.count("\n"), 0

    def thread_init(self):
        pass

    def pack(self, reads):
        return reads

    def set_reads(self, h, packed):
        h.reads = packed

    def device_part(self, h):
        return h

    def format(self, h, nthreads):
        return "".join("read%d\t%s\n" % (i, r[:16]) for i, r in enumerate(h.reads)).encode()

    def sync(self):
        pass


def cpu_legs(args, mode, gfa, reads, gpu_text_of, cores):
    """cpu_baseline legs on the host cores (oracle = CPU restatement, kind "port") + the in-run parity gate: every read
    a leg aligns is compared byte for byte with the GPU text of the same read.  Returns (cpu_baseline dict, checked,
    mismatches)."""
    from oracle import oracle as O
    og = O.Graph.from_gfa_text(gfa)
    omode = {0: O.M0_SIMD, 2: O.M2, 4: O.M4_ABS, 8: O.M8_ABS}[mode]
    what = {0: "oracle m0 (AVX2 semantics, scalar code)", 2: "oracle m2", 4: "oracle absolute-form m4",
            8: "oracle absolute-form m8 with the exact pruned search"}[mode]
    cap = args.cpu_reads if args.cpu_reads > 0 else 1 << 30
    checked, bad = 0, []
    pos = 0

    def leg(nreads, nthreads):
        nonlocal pos, checked
        nreads = max(1, min(nreads, cap, len(reads)))
        if pos + nreads > len(reads):
            pos = 0
        lo = pos
        pos += nreads
        secs, _, texts = og.bench_text(omode, reads[lo:lo + nreads], nthreads=nthreads, name_prefix="read", idx_base=1 + lo)
        # oracle names are read<k> with k relative to the slice: rebuild the GPU names the same way
        for k, t in enumerate(texts):
            exp = gpu_text_of(lo + k, "read%d" % k, 1 + lo + k)
            checked += 1
            if t != exp and len(bad) < 5:
                bad.append({"read": lo + k, "cpu": t[-120:].decode(errors="replace"), "gpu": exp[-120:].decode(errors="replace")})
        return nreads / secs, nreads, secs

    per_read = {0: 0.002, 2: 0.01, 4: 0.25, 8: 0.8}[mode]           # rough single-thread seconds per read (sizing only)
    v1, n1, s1 = leg(max(4, int(6.0 / per_read)), 1)
    sweep = []
    tried = sorted({max(1, cores // 8), max(1, cores // 4), max(1, cores // 2), cores})
    for T in tried:
        if T == 1:
            sweep.append({"threads": 1, "reads_per_s": round(v1, 3), "reads": n1, "secs": round(s1, 2)})
            continue
        v, n, s = leg(T * max(2, int(2.0 / per_read) if per_read < 0.1 else 2), T)
        sweep.append({"threads": T, "reads_per_s": round(v, 3), "reads": n, "secs": round(s, 2)})
    best = max(sweep, key=lambda e: e["reads_per_s"])
    cpu = {"value": best["reads_per_s"], "unit": "reads/s", "cores": best["threads"], "kind": "port",
           "sample": "%s; reads of the last timed step; best of a thread-count sweep %s on %d host cores (%d reads, %.1f s)"
                     % (what, [e["threads"] for e in sweep], cores, best["reads"], best["secs"]),
           "single_thread": {"value": round(v1, 4), "unit": "reads/s", "reads": n1, "secs": round(s1, 2)},
           "all_cores": {"value": best["reads_per_s"], "unit": "reads/s", "threads": best["threads"]},
           "thread_sweep": sweep}
    if mode == 8:
        # FAITHFUL figure: literal transliteration with the UNPRUNED O(L^2 n) best_alignment scan
        # (pathwise_alignment_recombination.rs:808-864).  DP timed in full, the scan on every `stride`-th column and
        # scaled to all columns — an extrapolation, said so here.
        T = max(1, min(8, cores, cap))
        stride = 100
        f = og.bench_faithful(reads[:T], nthreads=T, col_stride=stride)
        per = (f["dp_secs"] + f["scan_secs"] * f["cols_total"] / max(1, f["cols_visited"])) / T
        cpu["faithful_extrapolated"] = {
            "value": round(1.0 / per, 5), "unit": "reads/s per thread", "all_cores_if_linear": round(cores / per, 3),
            "reads": T, "threads": T, "dp_secs_per_read": round(f["dp_secs"] / T, 2),
            "scan_secs_per_read_extrapolated": round(f["scan_secs"] * f["cols_total"] / max(1, f["cols_visited"]) / T, 2),
            "note": "literal DP timed in full; unpruned scan timed on %d of %d columns per read and scaled"
                    % (f["cols_visited"] // T, f["cols_total"] // T)}
    return cpu, checked, bad


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))                       # before anything in this process touches the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (args.gpus, world))
        sys.exit(2)
    stub = os.environ.get("RG_BENCH_STUB") == "1"
    import torch
    import torch.distributed as dist
    dist_on = world > 1
    if not stub and torch.cuda.device_count() <= local_rank:
        sys.stderr.write("bench.py: rank %d needs GPU %d, %d visible\n" % (rank, local_rank, torch.cuda.device_count()))
        sys.exit(2)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if stub:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from recgraph_amd import synth
    from recgraph_amd.shard import gather_text, shard_bounds

    cfg = synth.CONFIGS[args.config]
    mode = cfg["mode"]
    num = int(args.config[1])
    batch = args.batch or DEFAULT_BATCH[args.config]
    sg, _, _ = synth.make_config(args.config, n_reads=1)
    gfa = sg.gfa()

    def make_reads(n, seed):
        if args.config in ("C2", "C3"):
            return synth.substring_reads(sg, n, cfg["n"], seed=seed)
        return synth.haplotype_reads(sg, n, cfg["n"], seed=seed, mosaic_frac=0.5 if args.config == "C5" else 0.0)

    # read set: `nb` distinct batches.  weak: seeded per rank.  strong: the N=1 read set, this rank's shard of every batch.
    nb = min(args.steps, DISTINCT_BATCHES)
    if args.scaling == "weak":
        batches = [make_reads(batch, 5678 + num + 1000 * rank + 100000 * (i + 1)) for i in range(nb)]
    else:
        lo, hi = shard_bounds(batch, rank, world)
        batches = [make_reads(batch, 5678 + num + 100000 * (i + 1))[lo:hi] for i in range(nb)]
    warm = make_reads(len(batches[0]), 5678 + num + 1000 * rank)
    nreads_step = len(batches[0])

    dev = local_rank
    eng = (StubEngine if stub else HipEngine)(dev, gfa, mode, warm, max(1, args.handles))
    hs = eng.handles
    batch_reads = batches                       # strings: the parity gate and the CPU legs align these
    batches = [eng.pack(b) for b in batches]    # the C ABI's input form, built once (not part of the hot path)
    warm_packed = eng.pack(warm)
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(args.device_threads, initializer=eng.thread_init)
    cores = os.cpu_count() or 1
    nthreads = max(1, min(16, cores // max(1, min(world, 8))))
    kstats = {}
    cells_total = 0

    def run_steps(k, read_sets, record, conc=None, stats=None):
        """k steps over `read_sets` (cycled).  Per step and handle: set_reads (host canonicalisation + H2D) -> device part
        (kernels + record fetch, a device thread) -> format (host threads).  With two handles the device part of step
        i+1 overlaps the formatting of step i and the set_reads of step i+2 (both on the main thread); with
        --device-threads 2 the device parts of the two handles are also submitted concurrently (two streams): the
        latency-bound small kernels of one handle then run beside the sweeps of the other."""
        nonlocal cells_total
        conc = (args.device_threads > 1) if conc is None else conc
        stats = kstats if stats is None else stats
        texts = []
        last = None
        nh = len(hs)
        futs = {}
        for j in range(min(nh, k)):
            eng.set_reads(hs[j], read_sets[j % len(read_sets)])
            if j == 0 or conc:
                futs[j] = pool.submit(eng.device_part, hs[j])
        for i in range(k):
            tw = time.perf_counter()
            cur = futs.pop(i).result()
            eng.host_s["wait_for_device"] = eng.host_s.get("wait_for_device", 0.0) + time.perf_counter() - tw
            if i + 1 < k and nh > 1 and (i + 1) not in futs:
                futs[i + 1] = pool.submit(eng.device_part, hs[(i + 1) % nh])
            if record:
                for kk, (ms, nl) in cur.kernel_stats().items():      # before the handle is reused
                    acc = stats.setdefault(kk, [0.0, 0])
                    acc[0] += ms
                    acc[1] += nl
                if stats is kstats:
                    cells_total += cur.cell_updates
            tf = time.perf_counter()
            texts.append(eng.format(cur, nthreads))
            eng.host_s["format"] = eng.host_s.get("format", 0.0) + time.perf_counter() - tf
            last = (cur, i % len(read_sets))
            if i + nh < k:
                eng.set_reads(cur, read_sets[(i + nh) % len(read_sets)])
                if conc or nh == 1:
                    futs[i + nh] = pool.submit(eng.device_part, cur)
        return texts, last

    if args.warmup:
        run_steps(args.warmup, [warm_packed], False)

    def sync():
        eng.sync()
        if dist_on:
            dist.barrier()
            eng.sync()

    sync()
    eng.host_s.clear()
    t0 = time.perf_counter()
    texts, (last_h, last_set) = run_steps(args.steps, batches, True)
    # final gather of ALL the GAF records of the timed steps to rank 0 (RCCL over xGMI when N > 1)
    parts = gather_text(b"".join(texts), rank, world, device="cpu" if (stub or not dist_on) else "cuda")
    gathered_bytes = sum(len(x) for x in parts) if parts is not None else 0
    sync()
    dt = time.perf_counter() - t0
    if dist_on:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if stub else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        ct = torch.tensor([float(cells_total), float(nreads_step)], dtype=torch.float64,
                          device="cpu" if stub else "cuda")
        dist.all_reduce(ct, op=dist.ReduceOp.SUM)
        cells_all, reads_step_all = float(ct[0].item()), int(ct[1].item())
    else:
        cells_all, reads_step_all = float(cells_total), nreads_step
    # Per-kernel durations for the roofline: with two device threads the kernels of the two handles share the GPU, so a
    # kernel's HIP-event time in the timed region includes the other stream's kernels.  Two probe steps with the device
    # parts serialised (after the timed region, same read sets) give the kernel's own duration.
    probe = {}
    if args.device_threads > 1 and len(hs) > 1 and rank == 0:
        saved = dict(eng.host_s)
        _, (last_h, last_set) = run_steps(min(2, args.steps), batches, True, conc=False, stats=probe)   # (the parity gate then checks the probe's last step)
        eng.host_s.clear()
        eng.host_s.update(saved)
    rc = 0
    if rank == 0:
        total_reads = reads_step_all * args.steps
        kroof = probe if probe else kstats
        probe_steps = min(2, args.steps) if probe else args.steps
        sweeps = {k: v for k, v in kroof.items() if k.startswith(("k_sweep", "k_m0", "k_m2"))}
        roof = None
        use16 = any(k.startswith("k_sweep16") for k in kstats)
        if sweeps:
            # dominant kernel family: the DP sweep.  One launch sweeps the whole graph once for one chunk of the batch.
            ms = sum(v[0] for v in sweeps.values())
            launches = sum(v[1] for v in sweeps.values())
            counting = sum(v[1] for k, v in sweeps.items() if not k.endswith("_colmax")) or launches
            reads_per_launch = nreads_step * probe_steps * (2 if mode == 8 else 1) / counting if mode in (4, 8) else nreads_step * probe_steps / counting
            per_launch_units = cells_total / args.steps * probe_steps / counting   # cell-updates one sweep launch processes (this rank)
            avg_s = ms / launches / 1e3
            algo = per_launch_units * BYTES_PER_CELL_UPDATE[mode] / avg_s / 1e9
            kname = {0: "k_m0_simd", 2: "k_poa_banded<true>", 4: "k_sweep", 8: "k_sweep"}[mode] + ("16" if use16 else "")
            roof = {"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                    "kernel": kname, "avg_launch_ms": round(ms / launches, 3), "launches": launches,
                    "reads_per_launch": round(reads_per_launch, 1),
                    "durations_from": ("%d probe steps with the device parts serialised, after the timed region (the timed "
                                       "steps run several handles concurrently: kernel_ms_per_step includes the other streams)" % probe_steps)
                    if probe else "the timed region (HIP events on the batch stream)",
                    # SURVEY §8d figure (the reference's own L x (n+1) x P matrices): NOT a fraction of anything this
                    # design moves — rows stay packed in registers / cache, so it exceeds the HBM peak by construction
                    "algorithmic_equiv_GBps": round(algo, 1)}
            cj = os.path.join(ROOT, "profiles", "counters_%s.json" % args.config)
            vj = os.path.join(ROOT, "profiles", "valu_calib.json")
            if os.path.exists(cj):
                try:
                    c = json.load(open(cj))
                    roof["counters_from"] = c.get("source")
                    if c.get("kernel_base") != kname:
                        roof["counters_stale"] = "profiled kernel %s, running %s" % (c.get("kernel_base"), kname)
                    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, calibrated units: see the file), per read
                    # per launch, scaled to this run's reads per launch; fabric-side bytes (Infinity-Cache hits included)
                    traffic = c["hbm_bytes_per_read_per_launch"] * reads_per_launch
                    hbm = {"achieved": round(traffic / avg_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(traffic / avg_s / 1e9 / HBM_PEAK_GBS, 4)}
                    roof["traffic"] = round(traffic)
                    roof["hbm"] = hbm
                    cand = [("hbm", hbm)]
                    if os.path.exists(vj) and c.get("valu_winstr_per_read_per_launch"):
                        peak = json.load(open(vj))["peak_winstr_per_s"]
                        rate = c["valu_winstr_per_read_per_launch"] * reads_per_launch / avg_s
                        valu = {"achieved": round(rate / 1e9, 2), "peak": round(peak / 1e9, 2), "unit": "G wave-instr/s",
                                "frac": round(rate / peak, 4)}
                        roof["valu"] = valu
                        cand.append(("valu", valu))
                    b, top = max(cand, key=lambda kv: kv[1]["frac"])
                    roof.update(bound=b, achieved=top["achieved"], peak=top["peak"], unit=top["unit"], frac=top["frac"])
                except Exception as ex:      # a broken counters file must not invalidate the throughput line
                    roof["counters_error"] = repr(ex)
        out = {
            "metric": "aligned reads/sec (-m 8 recombination, 1 kbp reads, 10k-row/32-path graph)" if args.config == "C5"
            else "aligned reads/sec (%s)" % args.config,
            "value": round(total_reads / dt, 2), "unit": "reads/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None,
            # arithmetic type of the DP cells: packed 16-bit integers when the batch's scores provably fit, else int32
            "dtype": "int16" if use16 else "int32", "data": "synthetic" if not stub else "STUB: no device work (RG_BENCH_STUB=1)",
            "config": {"workload": "BASELINE.json configs[%d] (%s): -m %d, %d bp reads, graph rows=%d paths=%d, "
                                   "%d reads/step over all GPUs, %d distinct reads timed, reads uploaded inside every step"
                       % (num - 1, args.config, mode, cfg["n"], eng.rows, eng.paths, reads_step_all,
                          reads_step_all * min(args.steps, nb)),
                       "parallelism": "read-shard x%d" % world},
            "cell_updates_per_s": round(cells_all / dt, 1),
            # HIP-event time per kernel and step.  With several device threads the steps overlap on the GPU: the figures of
            # the timed region include the other streams' kernels (their sum exceeds the step), the probe figures are
            # the kernels' own durations (their sum is the GPU time one step would take alone)
            "kernel_ms_per_step": {k: round(v[0] / probe_steps, 3) for k, v in kroof.items()},
            "kernel_ms_per_step_in_timed_region": {k: round(v[0] / args.steps, 3) for k, v in kstats.items()} if probe else None,
            # host wall time per step: device thread = run + fetch, main thread = format + set_reads (canonicalise + upload
            # of the batch two steps ahead) + wait_for_device; the two threads overlap
            "host_ms_per_step": {k: round(v / args.steps * 1e3, 3) for k, v in eng.host_s.items()},
            "gaf_bytes_gathered": gathered_bytes,
            "roofline": roof,
        }
        cpu = None
        if not args.no_cpu and not stub:
            reads_last = batch_reads[last_set]
            gpu_cores = cores if world == 1 else max(1, cores // world)
            if world == 1 and args.cpu_reads != 0:
                cpu, checked, bad = cpu_legs(args, mode, gfa, reads_last, lambda i, nm, idx: last_h.gaf_text(i, nm, idx).encode(), gpu_cores)
            else:
                # N > 1 (or --cpu-reads 0): parity gate only, on a small sample
                from oracle import oracle as O
                og = O.Graph.from_gfa_text(gfa)
                omode = {0: O.M0_SIMD, 2: O.M2, 4: O.M4_ABS, 8: O.M8_ABS}[mode]
                n = min(len(reads_last), max(8, min(64, gpu_cores * 2)))
                _, _, ts = og.bench_text(omode, reads_last[:n], nthreads=min(gpu_cores, n))
                checked, bad = n, []
                for k, t in enumerate(ts):
                    exp = last_h.gaf_text(k, "read%d" % k, 1 + k).encode()
                    if t != exp and len(bad) < 5:
                        bad.append({"read": k, "cpu": t[-120:].decode(errors="replace"), "gpu": exp[-120:].decode(errors="replace")})
            out["parity_checked"] = checked
            out["parity_ok"] = not bad
            if bad:
                out["parity_mismatches"] = bad
                rc = 3
        out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
