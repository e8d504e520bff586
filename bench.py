#!/usr/bin/env python3
"""bench.py — headline benchmark of the RecGraph DP hot path on MI355X.

Workload (BASELINE.json configs[4], the configuration the north-star target is quoted on): -m 8 recombination
alignment (R=4 r=0.1 B=1) of synthetic 1 kbp reads against a fixed synthetic ~10 k-row / 32-path graph.  The read set is
25 distinct batches of 4096 reads (102 400 distinct reads per GPU, seeded).  A "step" is one pass of the hot path over ONE
batch = one tile of the library's streaming engine (rg_stream_*, include/recgraph_hip.h): the timed region starts with
the graph resident and the reads in host memory in the C ABI's input form (bases + offsets), pushes every step's batch
into ONE rg_stream and takes the steps' GAF text back in input order.  Everything else happens behind that boundary:
upload of the batch (4 MB per step, PCIe-inclusive), two DP sweeps, candidate expansion, search, layer rebuild and
traceback on the device, record fetch and GAF formatting on the host, `--handles` batch handles (work-buffer sets in
HBM, each with its own host thread and HIP stream) pulling tiles from one queue so that the latency-bound small kernels of
one step fill the gaps of the other steps' sweeps.  This is the same call sequence `python -m recgraph_amd.cli` and the
plain-C caller (tests/c/abi_smoke.c) make.  The GAF text of EVERY timed step is gathered to rank 0 inside the timed
region (N > 1: over RCCL, step by step on a side thread while later steps run).

Multi-GPU: reads shard across ranks (one process per GPU, graph replicated, no data-path collective).  Launched by
torchrun (RANK / LOCAL_RANK / WORLD_SIZE in the environment) or directly: `python bench.py --gpus N` starts the N rank
processes itself before anything touches the GPU.  `--scaling weak` (default): every rank aligns steps x batch reads of
its own.  `--scaling strong`: the steps x batch reads of the N=1 run are split EVENLY over the ranks (reads, not tiles:
12 800 per rank for 102 400 on 8 GPUs), each share cut into even tiles of at most `batch` reads.  BASELINE.json's target
is the strong one ("100k reads ... >= 6x at 8 GPUs") and the driver cannot pass flags, so a default N > 1 run times BOTH:
`value` is the weak aggregate, `strong_100k` the 102 400-read set sharded over the N GPUs; at N = 1 `strong_proxy` times
the share one rank would get at N = 8 (12 800 reads) against the full set on the same GPU — the ratio bounds the
strong-scaling efficiency.

After the timed region the ranks leave the process group; rank 0 alone then measures the kernels' own durations (probe
steps on a one-handle stream), compares the GAF text of reads of the last timed step byte for byte with the CPU
restatement's (the in-run parity gate: exit status 3 and "parity_ok": false on any difference) and times the CPU legs of
`cpu_baseline` on worker PROCESSES that were forked at the very top of main(), before this process touched the GPU.
Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import pickle
import struct
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_CELL_UPDATE = {0: 12, 2: 32, 4: 8, 8: 12}   # SURVEY §8d algorithmic bytes per unit of work
HBM_PEAK_GBS = 8000.0                                 # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_NOMINAL_CYCLES = 2.0                             # MI355X_MICROARCH.md: one wave64 VALU instruction per SIMD every 2 cycles
DISTINCT_BATCHES = 25                                 # 25 x 4096 = 102 400 distinct reads per GPU
# Tiles of a rank's share in the strong regions: a short sequence of tiles (12 800 reads per rank at N = 8) ends with its
# last tiles running alone on the GPU; smaller tiles keep every handle busy to the end (measured: profiles/r04_notes.md)
STRONG_TILE = {"C4": 4096, "C5": 4096}
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
    ap.add_argument("--handles", type=int, default=0, help="batch handles (work-buffer sets in HBM, host thread + HIP stream each) of the stream; default: 3, config 4: 6 "
                    "(its sweep runs three waves per SIMD and leaves the small kernels of the other tiles little room: with six tiles in flight "
                    "the stream is steady at ~300 k reads/s, with three it flips between 270 k and 298 k from run to run)")
    ap.add_argument("--no-probe", action="store_true", help="skip the probe steps (kernel durations then come from the timed region)")
    ap.add_argument("--sweep-i32", action="store_true", help="force the i32 sweep kernel (rg_set_option sweep_i32)")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong_100k region (N > 1) / the strong_proxy region (N = 1)")
    ap.add_argument("--strong-ramp", type=int, default=1, help="first tiles of a share cut in two, a short one first (shard.even_tiles): the handles start out of phase, so the fill of a short tile sequence costs less (12 800 reads on one GPU: 126 ms with one ramp tile against 139 ms without, profiles/r05_notes.md)")
    ap.add_argument("--strong-tile", type=int, default=0, help="largest tile of a rank's share in the strong regions (default: see STRONG_TILE)")
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


# ----------------------------------------------------------------------------------------------------------------------
# CPU legs: worker processes forked BEFORE this process imports torch or touches the GPU (one per host CPU, pinned).
# Threads of one process share an address space: the restatement's per-read work buffers (40 MB vectors: mmap / munmap,
# page faults, TLB shootdowns to every core running a thread of the process) stop scaling at a few dozen threads;
# processes do not share any of that.
class CpuPool:
    def __init__(self, n):
        self.workers = []
        try:
            cpus = sorted(os.sched_getaffinity(0))
        except AttributeError:
            cpus = list(range(n))
        for i in range(n):
            p2c_r, p2c_w = os.pipe()
            c2p_r, c2p_w = os.pipe()
            pid = os.fork()
            if pid == 0:
                try:
                    os.close(p2c_w)
                    os.close(c2p_r)
                    for w in self.workers:
                        os.close(w[1])
                        os.close(w[2])
                    try:
                        os.sched_setaffinity(0, {cpus[i % len(cpus)]})
                    except Exception:
                        pass
                    _cpu_worker(p2c_r, c2p_w)
                finally:
                    os._exit(0)
            os.close(p2c_r)
            os.close(c2p_w)
            self.workers.append((pid, p2c_w, c2p_r))

    @staticmethod
    def _send(fd, obj):
        b = pickle.dumps(obj)
        os.write(fd, struct.pack("<q", len(b)))
        off = 0
        while off < len(b):
            off += os.write(fd, b[off:off + (1 << 16)])

    @staticmethod
    def _recv(fd):
        def rd(n):
            buf = b""
            while len(buf) < n:
                c = os.read(fd, n - len(buf))
                if not c:
                    raise RuntimeError("cpu worker died")
                buf += c
            return buf
        n = struct.unpack("<q", rd(8))[0]
        return pickle.loads(rd(n))

    def call(self, idx, obj):
        for i in idx:
            self._send(self.workers[i][1], obj)
        return [self._recv(self.workers[i][2]) for i in idx]

    def scatter(self, idx, objs):
        for i, o in zip(idx, objs):
            self._send(self.workers[i][1], o)
        return [self._recv(self.workers[i][2]) for i in idx]

    def close(self):
        for pid, w, r in self.workers:
            try:
                self._send(w, ("quit",))
            except Exception:
                pass
        for pid, w, r in self.workers:
            try:
                os.waitpid(pid, 0)
            except Exception:
                pass
            os.close(w)
            os.close(r)
        self.workers = []


def _cpu_worker(rfd, wfd):
    """Child: ("load", gfa, omode) -> graph built; ("reads", reads, name_base, idx_base) -> stored; ("go",) -> aligned
    on ONE thread, returns (seconds, [text per read], minor page faults); ("faithful", read, stride) -> probe dict."""
    import resource
    from oracle import oracle as O
    O.use_native()            # built by the parent before the fork (see main): loading only
    og = None
    omode = None
    job = None
    while True:
        msg = CpuPool._recv(rfd)
        if msg[0] == "quit":
            return
        if msg[0] == "load":
            og = O.Graph.from_gfa_text(msg[1])
            omode = getattr(O, msg[2])
            CpuPool._send(wfd, "ok")
        elif msg[0] == "reads":
            job = msg[1:]
            CpuPool._send(wfd, "ok")
        elif msg[0] == "go":
            reads, name_base, idx_base = job
            f0 = resource.getrusage(resource.RUSAGE_SELF).ru_minflt
            secs, _, texts = og.bench_text(omode, reads, nthreads=1, name_prefix="read", name_base=name_base, idx_base=idx_base)
            CpuPool._send(wfd, (secs, texts, resource.getrusage(resource.RUSAGE_SELF).ru_minflt - f0))
        elif msg[0] == "faithful":
            CpuPool._send(wfd, og.bench_faithful([msg[1]], nthreads=1, col_stride=msg[2]))


def cpu_legs(args, mode, gfa, reads, first, gpu_text_of, cores, pool):
    """cpu_baseline legs on the host cores (oracle = CPU restatement, kind "port") + the in-run parity gate: every read a
    leg aligns is compared byte for byte with the GPU text of the same read.  `first`: stream index of reads[0] (the
    stream names read i "read<i>").  Returns (cpu_baseline dict, checked, mismatches)."""
    omode = {0: "M0_SIMD", 2: "M2", 4: "M4_ABS", 8: "M8_ABS"}[mode]
    what = {0: "oracle m0 (AVX2 semantics, scalar code)", 2: "oracle m2", 4: "oracle absolute-form m4",
            8: "oracle absolute-form m8 with the exact pruned search"}[mode]
    cap = args.cpu_reads if args.cpu_reads > 0 else 1 << 30
    checked, bad = 0, []
    pos = 0
    nw = len(pool.workers)
    pool.call(range(nw), ("load", gfa, omode))

    def leg(per_worker, T):
        """T worker processes, `per_worker` reads each, started together; wall time from the first 'go' to the last result."""
        nonlocal pos, checked
        per_worker = max(1, min(per_worker, cap // T if cap >= T else 1, len(reads) // T))
        need = per_worker * T
        if pos + need > len(reads):
            pos = 0
        lo = pos
        pos += need
        idx = list(range(T))
        pool.scatter(idx, [("reads", reads[lo + k * per_worker:lo + (k + 1) * per_worker], first + lo + k * per_worker,
                            1 + first + lo + k * per_worker) for k in range(T)])
        t0 = time.perf_counter()
        res = pool.call(idx, ("go",))
        wall = time.perf_counter() - t0
        faults = 0
        for k, (secs, texts, flt) in enumerate(res):
            faults += flt
            for j, t in enumerate(texts):
                i = lo + k * per_worker + j
                exp = gpu_text_of(i)
                checked += 1
                if t != exp and len(bad) < 5:
                    bad.append({"read": i, "cpu": t[-120:].decode(errors="replace"), "gpu": exp[-120:].decode(errors="replace")})
        return need / wall, need, wall, max(r[0] for r in res), faults / need

    per_read = {0: 0.002, 2: 0.01, 4: 0.12, 8: 0.35}[mode]           # rough single-thread seconds per read (sizing only)
    v1, n1, s1, _, f1 = leg(max(4, int(5.0 / per_read)), 1)
    sweep = [{"threads": 1, "reads_per_s": round(v1, 3), "reads": n1, "secs": round(s1, 2), "per_thread_efficiency": 1.0,
              "minor_faults_per_read": round(f1)}]
    for T in sorted({max(1, cores // 4), max(1, cores // 2), cores, nw}):
        if T == 1 or T > nw:
            continue
        v, n, s, slowest, flt = leg(max(2, int(3.0 / per_read)), T)
        sweep.append({"threads": T, "reads_per_s": round(v, 3), "reads": n, "secs": round(s, 2),
                      "per_thread_efficiency": round(v / T / v1, 3), "slowest_worker_secs": round(slowest, 2),
                      "minor_faults_per_read": round(flt)})
    # the baseline is the leg with one worker per USABLE CPU; a leg beyond that (2x: shown so that the quota is visible) never
    # stands for the host, whatever noise does to it
    full = [e for e in sweep if e["threads"] == min(cores, nw)] or [max(sweep, key=lambda e: e["threads"])]
    best = full[0]
    cpu = {"value": best["reads_per_s"], "unit": "reads/s", "cores": best["threads"], "kind": "port",
           "sample": "%s; reads of the last timed step; %d single-threaded worker processes (forked before the GPU was touched, "
                     "pinned to distinct CPUs) = the %d usable host CPUs (%d reads, %.1f s); thread_sweep has the other legs %s"
                     % (what, best["threads"], cores, best["reads"], best["secs"], [e["threads"] for e in sweep]),
           "single_thread": {"value": round(v1, 4), "unit": "reads/s", "reads": n1, "secs": round(s1, 2)},
           "all_cores": {"value": best["reads_per_s"], "unit": "reads/s", "threads": best["threads"]},
           "thread_sweep": sweep}
    if mode == 8:
        # FAITHFUL figure: literal transliteration with the UNPRUNED O(L^2 n) best_alignment scan
        # (pathwise_alignment_recombination.rs:808-864).  DP timed in full, the scan on every `stride`-th column and
        # scaled to all columns — an extrapolation, said so here.
        T = max(1, min(8, nw, cap))
        stride = 100
        fs = pool.scatter(list(range(T)), [("faithful", reads[k], stride) for k in range(T)])
        dp = sum(f["dp_secs"] for f in fs)
        scan = sum(f["scan_secs"] for f in fs)
        cv, ct = sum(f["cols_visited"] for f in fs), sum(f["cols_total"] for f in fs)
        per = (dp + scan * ct / max(1, cv)) / T
        cpu["faithful_extrapolated"] = {
            "value": round(1.0 / per, 5), "unit": "reads/s per thread", "all_cores_if_linear": round(cores / per, 3),
            "reads": T, "threads": T, "dp_secs_per_read": round(dp / T, 2),
            "scan_secs_per_read_extrapolated": round(scan * ct / max(1, cv) / T, 2),
            "note": "literal DP timed in full; unpruned scan timed on %d of %d columns per read and scaled" % (cv // T, ct // T)}
    return cpu, checked, bad


# ----------------------------------------------------------------------------------------------------------------------
class StubStream:
    """RG_BENCH_STUB=1: NO device work — lets the CPU test suite run this file's launcher, sharding, barrier/all-reduce
    timing and per-step text gather (the world > 1 code path) over gloo.  Its JSON line says so; it is not a measurement.
    RG_BENCH_STUB_TILE_MS / RG_BENCH_STUB_CPU_MS / RG_BENCH_STUB_TEXT make it a HOST-SIDE model of one rank (VERDICT r4 #7:
    does an 8-rank node hold under a CPU quota?): a "device" thread that sleeps TILE_MS per tile (tiles one after the other,
    like the sweeps of one GPU), a "format" thread that burns CPU_MS of CPU per tile (what set_reads + fetch + the GAF
    formatter cost on the real stream), and TEXT bytes of record text per read for the gather."""
    handles = 0

    def __init__(self):
        self.q = []
        self.pos = 0
        self.tile_ms = float(os.environ.get("RG_BENCH_STUB_TILE_MS", "0"))
        self.cpu_ms = float(os.environ.get("RG_BENCH_STUB_CPU_MS", "0"))
        self.text = int(os.environ.get("RG_BENCH_STUB_TEXT", "16"))
        self.model = self.tile_ms > 0 or self.cpu_ms > 0
        if self.model:
            import hashlib
            import queue
            import threading
            burn = bytes(1 << 18)
            self.dev_q, self.fmt_q, self.done_q = queue.Queue(), queue.Queue(), queue.Queue()

            def device():
                while True:
                    it = self.dev_q.get()
                    if it is None:
                        self.fmt_q.put(None)
                        return
                    time.sleep(self.tile_ms / 1e3)
                    self.fmt_q.put(it)

            def fmt():
                while True:
                    it = self.fmt_q.get()
                    if it is None:
                        return
                    t_end = time.thread_time() + self.cpu_ms / 1e3
                    while time.thread_time() < t_end:      # busy for this thread's own CPU time — in C with the GIL released,
                        hashlib.sha256(burn).digest()      # like the product's C++ worker / formatter threads
                    self.done_q.put(it)
            self.threads = [threading.Thread(target=device, daemon=True), threading.Thread(target=fmt, daemon=True)]
            for th in self.threads:
                th.start()

    def _text(self, reads):
        if self.text == 16:
            return "".join("read%d\t%s\n" % (i, r[:16]) for i, r in enumerate(reads)).encode()
        return b"".join(b"read%d\t" % i + b"A" * self.text + b"\n" for i in range(len(reads)))

    def push(self, reads):
        item = (self.pos, reads)
        self.pos += len(reads)
        if self.model:
            self.dev_q.put(item)
        else:
            self.q.append(item)

    def next_text(self):
        first, reads = self.done_q.get() if self.model else self.q.pop(0)
        return first, len(reads), self._text(reads), None, 0

    def kernel_stats(self):
        return {}

    def close(self):
        if self.model:
            self.dev_q.put(None)


class HipStream:
    """The product: one rg_stream on this rank's GPU behind the C ABI (ctypes)."""

    def __init__(self, api, graph, params, dev, handles, tile, fmt_threads):
        self.api = api
        self.st = api.Stream(graph, params, device_ids=[dev], handles_per_device=handles, tile_reads=tile, format_threads=fmt_threads)

    def push(self, packed):
        self.st.push(packed)

    def next_text(self):
        t = self.st.next()
        return t.first, t.n, t.text, t, t.cell_updates

    def kernel_stats(self):
        return self.st.kernel_stats()

    @property
    def handles(self):
        return self.st.handles

    def close(self):
        self.st.close()


class StepGather:
    """Gather of every step's GAF text to rank 0, one step at a time on a side thread, while the main thread takes the next
    steps out of the stream (RCCL over xGMI on the GPU box, gloo in the CPU tests).  Steps are gathered in the same order
    on every rank; nothing else issues collectives while the thread runs."""

    def __init__(self, rank, world, device):
        self.rank, self.world, self.device = rank, world, device
        self.parts = []                # rank 0: per step, the list of per-rank payloads
        self.bytes = 0
        self.busy_s = 0.0
        self.host_waits = 0            # ranks > 0: times the fixed-capacity gather had to wait for a ring slot
        self.items = []
        self.cv = threading.Condition()
        self.done = False
        self.err = None
        self.th = None
        if world > 1:
            self.th = threading.Thread(target=self._run, daemon=True)
            self.th.start()

    def submit(self, data):
        if self.world == 1:
            self.parts.append([data])
            self.bytes += len(data)
            return
        with self.cv:
            self.items.append(data)
            self.cv.notify()

    def finish(self):
        """Blocks until every submitted step has been gathered; returns the seconds spent waiting here."""
        t0 = time.perf_counter()
        if self.th is not None:
            with self.cv:
                self.done = True
                self.cv.notify()
            self.th.join()
            if self.err is not None:
                raise self.err
        return time.perf_counter() - t0

    def _run(self):
        import torch
        from recgraph_amd.shard import FixedGather, gather_parts
        try:
            if self.device != "cpu":
                torch.cuda.set_device(self.device)
            fixed = None
            while True:
                with self.cv:
                    while not self.items and not self.done:
                        self.cv.wait()
                    if not self.items:
                        break
                    data = self.items.pop(0)
                t0 = time.perf_counter()
                if fixed is None:
                    # the first step goes through the variable-length gather — every rank learns every rank's size there —
                    # and sizes the fixed-capacity gather of all later steps (4 x the largest payload: a tile's text is
                    # within a few per cent of the next tile's): no host synchronisation on ranks > 0 from then on
                    parts, sizes = gather_parts(data, self.rank, self.world, self.device)
                    if self.rank == 0:
                        self.parts.append(parts)             # kept as tensors: no per-part bytes objects
                        self.bytes += sum(sizes)
                    # (RG_BENCH_VARIABLE_GATHER=1 keeps the variable-length gather for every step; so does a FixedGather that
                    # cannot be set up — every rank runs the same code on the same sizes, so they all fall back together)
                    if os.environ.get("RG_BENCH_VARIABLE_GATHER", "0") not in ("", "0"):
                        fixed = False
                    else:
                        try:
                            fixed = FixedGather(self.rank, self.world, self.device, cap=4 * max(1024, max(sizes)))
                        except Exception as ex:
                            print("bench: FixedGather unavailable (%r): variable-length gather for every step" % (ex,), file=sys.stderr, flush=True)
                            fixed = False
                elif fixed is False:
                    parts, sizes = gather_parts(data, self.rank, self.world, self.device)
                    if self.rank == 0:
                        self.parts.append(parts)
                        self.bytes += sum(sizes)
                else:
                    fixed.submit(data)
                self.busy_s += time.perf_counter() - t0
            if fixed:
                t0 = time.perf_counter()
                res, sizes = fixed.finish()
                if self.rank == 0:
                    for pr, sz in zip(res, sizes):
                        self.parts.append(pr)
                        self.bytes += sum(sz)
                self.host_waits = fixed.host_waits
                self.busy_s += time.perf_counter() - t0
        except Exception as ex:      # surfaced by finish()
            self.err = ex


def cpu_limit():
    """(CPUs this process may use, description): the scheduler affinity capped by the cgroup CPU quota.  The GPU boxes
    of this pool show 256 CPUs and carry a quota of 16 (cpu.max "1600000 100000"): beyond 16 busy threads the container is
    throttled, which is what flattened the thread sweeps of earlier rounds at ~40 reads/s."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    why = "%d CPUs in the affinity mask" % n
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()])):
        try:
            q, per = parse(open(path).read())
            if q != "max" and int(q) > 0:
                lim = max(1, -(-int(q) // int(per)))
                if lim < n:
                    n, why = lim, "cgroup CPU quota %s/%s = %d CPUs (of %d visible)" % (q, per, lim, os.cpu_count() or 0)
            break
        except (OSError, ValueError):
            continue
    return n, why


_COMMON_SOURCES = ("rg_device.hpp", "rg_codes.hpp")
_PATH_SOURCES = ("rg_sweep16.hip", "rg_pathwise.hip", "rg_path_driver.hip", "rg_path_kernels.hpp", "rg_path_args.hpp") + _COMMON_SOURCES
KERNEL_SOURCES = {"C2": ("rg_poa.hip", "rg_poa_args.hpp") + _COMMON_SOURCES, "C3": ("rg_poa_banded.hip", "rg_poa_args.hpp") + _COMMON_SOURCES,
                  "C4": _PATH_SOURCES, "C5": _PATH_SOURCES}


def code_hash(config="C5"):
    """sha256 over the sources of the kernels a configuration runs (and, for the pathwise modes, the pipeline driver):
    profiles/counters_<config>.json carries the hash of the tree it was measured on (host-side files — ABI, stream, parsers —
    and the other modes' kernels do not change what its counters count)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "recgraph_amd", "csrc")
    for f in KERNEL_SOURCES[config]:
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def slice_packed(packed, lo, hi):
    """Reads [lo, hi) of a packed (bytes, int64 offsets) read set, as a packed set of its own."""
    blob, offs = packed
    return blob[int(offs[lo]):int(offs[hi])], (offs[lo:hi + 1] - offs[lo]).copy()


def blocking_sync(device):
    """N > 1: host waits of THIS process on the device sleep instead of spinning (hipDeviceScheduleBlockingSync, set before
    torch creates the context).  The per-step gather calls `.item()` / `.cpu()` on device tensors while the collective waits
    for its peers: spinning, that is up to one CPU per rank besides the stream's 0.6 — 8 ranks share a 16-CPU quota on the
    pool's boxes.  The library's own waits already sleep on events (rg_host.hpp).  Errors are ignored: the flag is
    an optimisation, RG_BENCH_SPIN_SYNC=1 leaves the default."""
    if os.environ.get("RG_BENCH_SPIN_SYNC") == "1":
        return False
    try:
        import ctypes
        import torch
        # THE runtime torch loaded (its wheel ships one; by soname the library's HIP calls resolve to the same copy) — a bare
        # "libamdhip64.so" would map the system's copy as a second runtime and set the flag there
        own = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        hip = ctypes.CDLL(own if os.path.exists(own) else "libamdhip64.so.7")
        if hip.hipSetDevice(ctypes.c_int(device)) != 0:
            return False
        return hip.hipSetDeviceFlags(ctypes.c_uint(0x4)) == 0        # hipDeviceScheduleBlockingSync
    except Exception:
        return False


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
    cores, cores_why = cpu_limit()
    # the CPU legs' worker processes: forked NOW, before torch / HIP are imported into this process
    pool = None
    want_cpu = rank == 0 and not args.no_cpu and not stub
    oracle_build = None
    if want_cpu:
        from oracle import oracle as O        # (ctypes only: nothing here touches the GPU)
        O.build()
        oracle_build = O.use_native() or "portable liboracle.so (g++ -O2): the native build failed"
    if want_cpu and world == 1 and args.cpu_reads != 0:
        pool = CpuPool(2 * cores if cores < (os.cpu_count() or 1) else cores)     # (one leg oversubscribes a quota: shown, not used)
    elif want_cpu:
        pool = CpuPool(max(1, min(8, cores // max(1, world))))      # parity gate only
    import resource
    import torch
    import torch.distributed as dist
    dist_on = world > 1
    if not stub and torch.cuda.device_count() <= local_rank:
        sys.stderr.write("bench.py: rank %d needs GPU %d, %d visible\n" % (rank, local_rank, torch.cuda.device_count()))
        sys.exit(2)
    if not stub and (dist_on or os.environ.get("RG_BENCH_BLOCKING_SYNC") == "1"):
        blocking_sync(local_rank)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if stub:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from recgraph_amd import synth
    from recgraph_amd.shard import even_tiles, shard_bounds

    cfg = synth.CONFIGS[args.config]
    mode = cfg["mode"]
    num = int(args.config[1])
    batch = args.batch or DEFAULT_BATCH[args.config]
    if args.handles <= 0:
        args.handles = 6 if args.config == "C4" else 3
    sg, _, _ = synth.make_config(args.config, n_reads=1)
    gfa = sg.gfa()

    def make_reads(n, seed):
        if args.config in ("C2", "C3"):
            return synth.substring_reads(sg, n, cfg["n"], seed=seed)
        return synth.haplotype_reads(sg, n, cfg["n"], seed=seed, mosaic_frac=0.5 if args.config == "C5" else 0.0)

    nb = min(args.steps, DISTINCT_BATCHES)
    # THE read set of the N = 1 run: batch i (of `batch` reads) has the seed below; the strong experiments shard exactly it
    canon_seed = lambda i: 5678 + num + 100000 * (i % nb + 1)
    _cache = {}

    def batch_strings(i, seed):
        if (i, seed) not in _cache:
            _cache[(i, seed)] = make_reads(batch, seed)
        return _cache[(i, seed)]

    dev = local_rank
    fmt_threads = max(1, min(16, cores // max(1, min(world, 8)) // max(1, args.handles)))
    if stub:
        main_stream = StubStream()
        pack = lambda reads: reads
        slice_set = lambda p, lo, hi: p[lo:hi]
        rows, paths = gfa.count("\n"), 0
        api = None
    else:
        from recgraph_amd import _lib, api
        _lib.check(_lib.load().rg_set_device(dev))
        if args.sweep_i32:
            api.set_option("sweep_i32", 1)
        graph = api.Graph.from_gfa_text(gfa)
        params = api.make_params(mode)
        rows, paths = graph.rows, graph.paths_number
        main_stream = HipStream(api, graph, params, dev, max(1, args.handles), batch, fmt_threads)
        pack = api.Batch.pack_reads          # the C ABI's input form, built once per read set (not part of the hot path)
        slice_set = slice_packed

    # (never above the stream's tile size: a larger push is cut in two by the engine and the region would time other tiles than it reports)
    strong_tile = min(batch, args.strong_tile or STRONG_TILE.get(args.config, batch))

    def share_tiles(total_batches, lo, hi, ramp=0, max_tile=None):
        """Reads [lo, hi) of the canonical read set of `total_batches` batches, as packed tiles of even size (<= batch):
        (tiles, [(batch index, first read in the batch, reads)] per tile for the parity gate)."""
        tiles, where = [], []
        pos = lo
        for sz in even_tiles(hi - lo, max_tile or batch, ramp):
            parts, w = [], []
            need = sz
            while need:
                bi, off = divmod(pos, batch)
                take = min(need, batch - off)
                parts.append(batch_strings(bi, canon_seed(bi))[off:off + take])
                w.append((bi, off, take))
                pos += take
                need -= take
            tiles.append(pack([r for p in parts for r in p]))
            where.append(w)
        return tiles, where

    # Read set of the headline region.
    #   weak: every rank aligns `steps` tiles of `batch` reads of its own (seeded per rank);
    #   strong: the steps x batch reads of the N = 1 run, split evenly over the ranks (reads, not tiles), even tiles
    if args.scaling == "weak":
        seed_of = lambda i: canon_seed(i) + 1000 * rank
        my_tiles = [pack(batch_strings(i % nb, seed_of(i))) for i in range(args.steps)] if nb == args.steps else None
        if my_tiles is None:
            distinct = {i: pack(batch_strings(i, seed_of(i))) for i in range(nb)}
            my_tiles = [distinct[i % nb] for i in range(args.steps)]
        last_strings = lambda: batch_strings((args.steps - 1) % nb, seed_of((args.steps - 1) % nb))
        tiles_per_rank = [args.steps] * world
    else:
        total = args.steps * batch
        spans = [shard_bounds(total, r, world) for r in range(world)]
        tiles_per_rank = [len(even_tiles(b - a, batch)) for a, b in spans]
        lo, hi = spans[rank]
        my_tiles, my_where = share_tiles(args.steps, lo, hi)
        last_strings = lambda: [r for bi, off, take in my_where[-1] for r in batch_strings(bi, canon_seed(bi))[off:off + take]] if my_where else []
    warm = pack(make_reads(batch, 5678 + num + 1000 * rank))

    def run_steps(stream, sets, gather=None):
        """Push every tile, then take them back in input order (each goes to the gather as it arrives)."""
        for t in sets:
            stream.push(t)
        last = None
        cells = [0, 0, 0]
        for _ in sets:
            first, n, text, tile, c = stream.next_text()
            cells[0] += c
            cells[1] += getattr(tile, "cell_updates_performed", 0) or 0
            st = getattr(tile, "status", None)
            if st is not None:
                cells[2] += int((st & 2).astype(bool).sum())       # RG_READ_BAND_NOT_ENOUGH: warning line + empty record
            if gather is not None:
                gather.submit(text)
            last = (first, n, text, tile)
        return last, cells

    # setup (neither timed nor warm-up): one tile per handle, so that every handle exists and owns its work buffers
    run_steps(main_stream, [warm] * max(1, args.handles))
    if args.warmup:
        run_steps(main_stream, [warm] * args.warmup)

    def sync():
        if not stub:
            torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            if not stub:
                torch.cuda.synchronize()

    def timed_region(tiles, pad_to):
        """Barrier + device sync, every tile of this rank through the stream with its text gathered to rank 0 as it arrives,
        gather complete, barrier + device sync; the time is the MAX over the ranks.  Returns a dict."""
        gather = StepGather(rank, world, "cpu" if (stub or not dist_on) else torch.device("cuda", local_rank))
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        sync()
        t0 = time.perf_counter()
        last, cells = run_steps(main_stream, tiles, gather)
        for _ in range(pad_to - len(tiles)):
            gather.submit(b"")             # every rank takes part in the same number of gathers
        gather_wait = gather.finish()      # the gather of ALL the GAF records of the timed tiles to rank 0 is complete here
        sync()
        dt = time.perf_counter() - t0
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
        nreads = sum(len(t[1]) - 1 if not stub else len(t) for t in tiles)
        res = {"dt_local": dt, "dt": dt, "last": last, "cells": float(cells[0]), "cells_perf": float(cells[1]), "band_not_enough": int(cells[2]), "reads": nreads,
               "reads_all": nreads, "gather_busy": gather.busy_s, "gather_wait": gather_wait, "bytes": gather.bytes,
               "cpu_s": cpu_s, "cpu_s_all": cpu_s}
        if dist_on:
            cdev = "cpu" if stub else "cuda"
            tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            res["dt"] = float(tt.item())
            ct = torch.tensor([res["cells"], float(nreads), res["cells_perf"], cpu_s], dtype=torch.float64, device=cdev)
            dist.all_reduce(ct, op=dist.ReduceOp.SUM)
            res["cells"], res["reads_all"], res["cells_perf"], res["cpu_s_all"] = float(ct[0].item()), int(ct[1].item()), float(ct[2].item()), float(ct[3].item())
            # per rank: CPU seconds of the region and the gather's ring waits (what an 8-GPU run asks of the host, rank by rank)
            pr = torch.tensor([cpu_s, float(gather.host_waits)], dtype=torch.float64, device=cdev)
            prs = [torch.zeros_like(pr) for _ in range(world)]
            dist.all_gather(prs, pr)
            res["cpu_s_per_rank"] = [float(x[0].item()) for x in prs]
            res["gather_ring_waits_per_rank"] = [int(x[1].item()) for x in prs]
        return res

    k0 = main_stream.kernel_stats()
    head = timed_region(my_tiles, max(tiles_per_rank))
    k1 = main_stream.kernel_stats()
    kstats = {k: (v[0] - k0.get(k, (0, 0))[0], v[1] - k0.get(k, (0, 0))[1]) for k, v in k1.items()}
    dt = head["dt"]
    last = head["last"]

    # The stated target (BASELINE.json: "100k reads sharded across the GPUs, >= 6x at 8") beside the headline, in the SAME
    # run because the driver passes no flags: the 25 x `batch` reads of the N = 1 run, split evenly over the ranks.
    strong = None
    if args.scaling == "weak" and dist_on and not args.no_strong:
        tb = DISTINCT_BATCHES
        spans = [shard_bounds(tb * batch, r, world) for r in range(world)]
        tpr = [len(even_tiles(b - a, strong_tile, args.strong_ramp)) for a, b in spans]
        s_tiles, _ = share_tiles(tb, *spans[rank], ramp=args.strong_ramp, max_tile=strong_tile)
        sres = timed_region(s_tiles, max(tpr))
        strong = {"reads": sres["reads_all"], "reads_per_s": round(sres["reads_all"] / sres["dt"], 2), "ms": round(sres["dt"] * 1e3, 2),
                  "reads_per_rank": [b - a for a, b in spans], "tiles_per_rank": tpr,
                  "tile_reads": [len(t[1]) - 1 if not stub else len(t) for t in s_tiles] if rank == 0 else None,
                  # against the weak per-GPU rate of this same run (the N = 1 rate is not measured in an N > 1 run)
                  "speedup_vs_weak_per_gpu_rate": round(sres["reads_all"] / sres["dt"] / (head["reads_all"] / dt / world), 3),
                  "host_cpu_s": round(sres["cpu_s_all"], 3)}
    # N = 1: what ONE rank sees at N = 8 (its 1/8 share of the 102 400 reads, even tiles) against the full set on this GPU:
    # fill / drain and tile-size effects are all of the strong-scaling loss there is (no data-path collective, graph
    # replicated), so ratio x 8 projects the 8-GPU speed-up.
    proxy = None
    if args.scaling == "weak" and not dist_on and not stub and not args.no_strong and args.config in ("C4", "C5"):
        share = DISTINCT_BATCHES * batch // 8
        p_tiles, _ = share_tiles(DISTINCT_BATCHES, 0, share, ramp=args.strong_ramp, max_tile=strong_tile)
        best = None
        for _ in range(2):                 # two passes, the better one (a single short region is noisy)
            pres = timed_region(p_tiles, len(p_tiles))
            best = pres if best is None or pres["dt"] < best["dt"] else best
        rate = share / best["dt"]
        proxy = {"reads": share, "tiles": [len(t[1]) - 1 for t in p_tiles], "ms": round(best["dt"] * 1e3, 2), "reads_per_s": round(rate, 1),
                 "ratio_vs_timed_region": round(rate / (head["reads_all"] / dt), 4),
                 "projected_speedup_at_8_gpus": round(8 * rate / (head["reads_all"] / dt), 2),
                 "note": "1/8 of the 102 400-read set on ONE GPU (what a rank aligns at N = 8) against the full timed region; no 8-GPU run behind it"}
    # POA configurations: the generator of SURVEY 8d draws substrings of a source->sink walk at a uniform offset, and a GLOBAL
    # alignment of such a read leaves the band (the reference prints "band not enough" and an empty record for nearly all of
    # them: `band_not_enough_fraction`), so the headline times the DP but hardly ever the traceback walker and the GAF
    # formatter on a real alignment.  `anchored`: the same shape with every read starting at the graph's source.
    anchored = None
    if mode in (0, 2) and not dist_on and not stub and not args.no_strong:
        def region(make, nt=3):
            tiles = [pack(make(k)) for k in range(nt)]
            run_steps(main_stream, tiles[:1])
            res = timed_region(tiles, len(tiles))
            return {"reads": res["reads_all"], "reads_per_s": round(res["reads_all"] / res["dt"], 1), "ms_per_tile": round(res["dt"] / len(tiles) * 1e3, 3),
                    "band_not_enough_fraction": round(res["band_not_enough"] / max(1, res["reads_all"]), 4)}
        anchored = {"source_anchored": region(lambda k: synth.substring_reads(sg, batch, cfg["n"], seed=777 + num + 31 * k, anchored=True)),
                    # whole source->sink walks (reads as long as the graph: ~%d bases): what a global alignment places inside the
                    # configuration's band, so the walker and the GAF formatter run on real alignments
                    "full_walks": region(lambda k: synth.full_walk_reads(sg, max(256, batch // 8), seed=991 + num + 31 * k)),
                    "note": "same stream, same parameters; `source_anchored`: reads of the configuration's length that start at the graph's "
                            "source; `full_walks`: whole source->sink walks"}
    if dist_on:
        # the ranks part here: rank 0's probe steps and CPU legs do not hold the other GPUs
        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:
            main_stream.close()
            sys.exit(0)
    cells_all, cells_perf_all, total_reads = head["cells"], head["cells_perf"], head["reads_all"]
    gathered_bytes = head["bytes"]
    handles_used = main_stream.handles
    main_stream.close()
    steps_here = max(1, len(my_tiles))

    # Per-kernel durations for the roofline: with several handles the kernels of concurrent tiles share the GPU, so a
    # kernel's HIP-event time in the timed region includes the other streams' kernels.  Probe steps on a ONE-handle stream
    # (after the timed region, same read sets) give the kernels' own durations.
    probe, probe_steps = {}, 0
    if not stub and not args.no_probe and args.handles > 1 and my_tiles:
        ps = HipStream(api, graph, params, dev, 1, batch, fmt_threads)
        run_steps(ps, [warm])
        p0 = ps.kernel_stats()
        probe_steps = min(2, len(my_tiles))
        run_steps(ps, my_tiles[:probe_steps])
        p1 = ps.kernel_stats()
        probe = {k: (v[0] - p0.get(k, (0, 0))[0], v[1] - p0.get(k, (0, 0))[1]) for k, v in p1.items()}
        ps.close()
    # The same pipeline with the i32 sweep forced (the reference's arithmetic type; the packed 16-bit rows are admitted by a
    # host-side range proof and hold the same integers): a few tiles on a stream of its own, after the timed region.
    int32 = None
    use16 = any(k.startswith("k_sweep16") for k in kstats)
    if not stub and use16 and not args.sweep_i32 and not args.no_probe and mode in (4, 8) and my_tiles:
        api.set_option("sweep_i32", 1)
        try:
            st32 = HipStream(api, graph, params, dev, max(1, args.handles), batch, fmt_threads)
            run_steps(st32, [warm] * max(1, args.handles))
            sets = (my_tiles * 3)[:3]
            torch.cuda.synchronize()
            t32 = time.perf_counter()
            run_steps(st32, sets)
            torch.cuda.synchronize()
            d32 = time.perf_counter() - t32
            n32 = sum(len(t[1]) - 1 for t in sets)
            ks = st32.kernel_stats()
            int32 = {"reads_per_s": round(n32 / d32, 1), "tiles": len(sets), "reads": n32, "ms_per_tile": round(d32 / len(sets) * 1e3, 2),
                     "kernels": sorted(k for k in ks if k.startswith("k_sweep"))}
            st32.close()
        finally:
            api.set_option("sweep_i32", 0)

    rc = 0
    kroof = probe if probe else kstats
    ksteps = probe_steps if probe else steps_here
    kern = {k: v for k, v in kroof.items() if not k.startswith(("host:", "mem:"))}
    # HBM work buffers per read as the path driver sized them (a pseudo-kernel: bytes summed per chunk | chunks)
    memstat = kstats.get("mem:work_bytes_per_read")
    hbm_per_read = int(memstat[0] / memstat[1]) if memstat and memstat[1] else None
    sweeps = {k: v for k, v in kern.items() if k.startswith(("k_sweep", "k_m0", "k_m2"))}
    roof = None
    if sweeps:
        # dominant kernel family: the DP sweep.  One launch sweeps the whole graph once for one chunk of the batch.
        ms = sum(v[0] for v in sweeps.values())
        launches = sum(v[1] for v in sweeps.values())
        counting = sum(v[1] for k, v in sweeps.items() if not k.endswith("_colmax")) or launches
        reads_probe = sum(len(t[1]) - 1 for t in my_tiles[:ksteps]) if probe else head["reads"]
        reads_per_launch = reads_probe * (2 if mode == 8 else 1) / counting if mode in (4, 8) else reads_probe / counting
        per_launch_units = head["cells"] / max(1, head["reads_all"]) * reads_probe / counting   # cell-updates one sweep launch stands for
        avg_s = ms / launches / 1e3
        algo = per_launch_units * BYTES_PER_CELL_UPDATE[mode] / avg_s / 1e9
        kname = {0: "k_m0_simd", 2: "k_poa_banded<true>", 4: "k_sweep", 8: "k_sweep"}[mode] + ("16" if use16 else "")
        roof = {"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                "kernel": kname, "avg_launch_ms": round(ms / launches, 3), "launches": launches,
                "reads_per_launch": round(reads_per_launch, 1),
                "durations_from": ("%d probe steps on a one-handle stream after the timed region (the timed steps run %d handles "
                                   "concurrently: their HIP-event times include the other streams' kernels)" % (probe_steps, handles_used))
                if probe else "the timed region (HIP events on the batch stream)",
                # SURVEY §8d figure (the reference's own L x (n+1) x P matrices, 12 B per member-cell update of the WORKLOAD):
                # NOT a fraction of anything this design moves or does — rows stay packed in registers / cache and gather
                # runs do not perform the member updates they stand for — so it exceeds the HBM peak by construction
                "algorithmic_equiv_GBps": round(algo, 1),
                "code_hash": code_hash(args.config)}   # of this configuration's kernel sources: counters collected on another tree are flagged stale
        cj = os.path.join(ROOT, "profiles", "counters_%s.json" % args.config)
        vj = os.path.join(ROOT, "profiles", "valu_calib.json")
        if os.path.exists(cj):
            try:
                c = json.load(open(cj))
                roof["counters_from"] = c.get("source")
                if c.get("kernel_base") != kname:
                    roof["counters_stale"] = "profiled kernel %s, running %s" % (c.get("kernel_base"), kname)
                elif c.get("code_hash") != roof["code_hash"]:
                    roof["counters_stale"] = "kernel sources changed since the counters were collected (%s -> %s)" % (c.get("code_hash"), roof["code_hash"])
                if "counters_stale" not in roof:
                    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, calibrated units: see the file), per read
                    # per launch, scaled to this run's reads per launch; fabric-side bytes (Infinity-Cache hits included)
                    traffic = c["hbm_bytes_per_read_per_launch"] * reads_per_launch
                    hbm = {"achieved": round(traffic / avg_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(traffic / avg_s / 1e9 / HBM_PEAK_GBS, 4)}
                    roof["traffic"] = round(traffic)
                    roof["hbm"] = hbm
                    cand = [("hbm", hbm)]
                    if os.path.exists(vj) and c.get("valu_winstr_per_read_per_launch"):
                        vc = json.load(open(vj))
                        peak = vc["peak_winstr_per_s"]
                        rate = c["valu_winstr_per_read_per_launch"] * reads_per_launch / avg_s
                        valu = {"achieved": round(rate / 1e9, 2), "peak": round(peak / 1e9, 2), "unit": "G wave-instr/s",
                                "frac": round(rate / peak, 4), "winstr_per_read_per_launch": round(c["valu_winstr_per_read_per_launch"])}
                        raw = vc.get("raw", {})
                        if raw.get("compute_units") and raw.get("clock_mhz"):
                            # the guide's nominal issue rate (one wave64 VALU instruction per SIMD every 2 cycles) beside the
                            # calibrated peak of this kernel's instruction mix
                            nominal = raw["compute_units"] * 4 * raw["clock_mhz"] * 1e6 / VALU_NOMINAL_CYCLES
                            valu["peak_nominal"] = round(nominal / 1e9, 2)
                            valu["frac_of_nominal"] = round(rate / nominal, 4)
                        # the same count on the STEP clock of the timed region: with several handles the sweeps of concurrent tiles
                        # overlap each other and the small kernels, so a step is not the sum of its lone launches (the probe clock
                        # above); sweeps only — the other kernels' instructions are not in this numerator
                        sweeps_per_read = 2 if mode == 8 else 1
                        step_rate = c["valu_winstr_per_read_per_launch"] * sweeps_per_read * head["reads"] / dt
                        valu["frac_step_clock"] = round(step_rate / peak, 4)
                        roof["clocks"] = {"probe_sweep_ms_per_step": round(ms / max(1, ksteps), 3), "timed_step_ms": round(dt / steps_here * 1e3, 3),
                                          "note": "frac: lone launches (probe steps, one handle); frac_step_clock: the timed region's own clock"}
                        roof["valu"] = valu
                        cand.append(("valu", valu))
                    b, top = max(cand, key=lambda kv: kv[1]["frac"])
                    roof.update(bound=b, achieved=top["achieved"], peak=top["peak"], unit=top["unit"], frac=top["frac"])
                # (stale counters: no fraction is printed — instruction and byte counts of another kernel build say nothing
                # about this one)
            except Exception as ex:      # a broken counters file must not invalidate the throughput line
                roof["counters_error"] = repr(ex)
    tile_sizes = sorted({len(t[1]) - 1 if not stub else len(t) for t in my_tiles})
    out = {
        "metric": "aligned reads/sec (-m 8 recombination, 1 kbp reads, 10k-row/32-path graph)" if args.config == "C5"
        else "aligned reads/sec (%s)" % args.config,
        "value": round(total_reads / dt, 2), "unit": "reads/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / max(1, args.steps) * 1e3, 3), "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None,
        # arithmetic type of the DP cells: packed 16-bit integers when the batch's scores provably fit, else int32
        "dtype": "int16" if use16 else "int32", "data": "synthetic" if not stub else "STUB: no device work (RG_BENCH_STUB=1)",
        "config": {"workload": "BASELINE.json configs[%d] (%s): -m %d, %d bp reads, graph rows=%d paths=%d, "
                               "%d reads/step over all GPUs, %d distinct reads timed, reads uploaded inside every step; "
                               "one rg_stream per GPU (%d handles), %s-read tiles"
                   % (num - 1, args.config, mode, cfg["n"], rows, paths,
                      batch * (world if args.scaling == "weak" else 1),
                      batch * min(args.steps, nb) * (world if args.scaling == "weak" else 1), handles_used,
                      "/".join(str(x) for x in tile_sizes) if tile_sizes else str(batch)),
                   "parallelism": "read-shard x%d" % world,
                   # what shapes the headline besides the workload: the defaults depend on the configuration
                   "handles": handles_used, "strong_ramp": args.strong_ramp},
        # member-row cell updates of the WORKLOAD per second (sum over rows of |paths(row)| x (n + 1), forward + reverse: the
        # reference's unit of work) ...
        "cell_updates_per_s": round(cells_all / dt, 1),
        # ... and the cell updates the kernels PERFORMED for it: a gather run of k_sweep16 does the alpha and a column map per
        # row and two passes per member and run instead of one update per member and row
        "cell_updates_performed_per_s": round(cells_perf_all / dt, 1) if cells_perf_all else None,
        # HIP-event time per kernel and step.  With several handles the steps overlap on the GPU: the figures of the timed
        # region include the other streams' kernels (their sum exceeds the step), the probe figures are the kernels' own
        # durations (their sum is the GPU time one step would take alone)
        "kernel_ms_per_step": {k: round(v[0] / max(1, ksteps), 3) for k, v in kern.items()},
        "kernel_ms_per_step_in_timed_region": {k: round(v[0] / steps_here, 3) for k, v in kstats.items() if not k.startswith(("host:", "mem:"))} if probe else None,
        # host wall time per step, summed over the stream's worker threads (they overlap each other and the device):
        # set_reads = canonicalise + upload, run = kernels (waiting for the device), fetch = records D2H, format = GAF text
        "host_ms_per_step": {k[5:]: round(v[0] / steps_here, 3) for k, v in kstats.items() if k.startswith("host:")},
        # CPU time (user + system, every thread of the rank processes: stream workers, formatting, gather, this loop) per step,
        # summed over the ranks: what an N-GPU run asks of the host
        "host_cpu_s_per_step": round(head["cpu_s_all"] / max(1, args.steps), 4),
        "host_cpus_busy": round(head["cpu_s_all"] / dt, 2),
        "host_cpus_busy_per_rank": [round(c / dt, 2) for c in head["cpu_s_per_rank"]] if "cpu_s_per_rank" in head else None,
        "gather_ring_waits_per_rank": head.get("gather_ring_waits_per_rank"),
        # work buffers in HBM per read of a chunk (pathwise modes; x tile reads x handles = what the stream holds)
        "hbm_work_bytes_per_read": hbm_per_read,
        "hbm_work_GB_held": round(hbm_per_read * batch * handles_used / 1e9, 2) if hbm_per_read else None,
        "tiles_per_rank": tiles_per_rank,
        "gather_ms_per_step": round((head["gather_busy"] if dist_on else 0.0) / steps_here * 1e3, 3),
        "gather_wait_ms": round(head["gather_wait"] * 1e3, 3),
        "gaf_bytes_gathered": gathered_bytes,
        "strong_100k": strong,
        "strong_proxy": proxy,
        "int32": int32,
        "roofline": roof,
    }
    if mode in (0, 2):
        out["band_not_enough_fraction"] = round(head["band_not_enough"] / max(1, head["reads"]), 4)
        out["anchored"] = anchored
    cpu = None
    if pool is not None:
        first, n_last, text_last, tile = last
        reads_last = last_strings()
        gpu_text_of = lambda i: tile.text_of(i)
        if world == 1 and args.cpu_reads != 0:
            cpu, checked, bad = cpu_legs(args, mode, gfa, reads_last, first, gpu_text_of, cores, pool)
        else:
            # N > 1 (or --cpu-reads 0): parity gate only, on a small sample
            omode = {0: "M0_SIMD", 2: "M2", 4: "M4_ABS", 8: "M8_ABS"}[mode]
            nw = len(pool.workers)
            pool.call(range(nw), ("load", gfa, omode))
            per = max(1, min(8, len(reads_last) // nw))
            pool.scatter(list(range(nw)), [("reads", reads_last[k * per:(k + 1) * per], first + k * per, 1 + first + k * per) for k in range(nw)])
            res = pool.call(list(range(nw)), ("go",))
            checked, bad = 0, []
            for k, (_, texts, _) in enumerate(res):
                for j, t in enumerate(texts):
                    exp = gpu_text_of(k * per + j)
                    checked += 1
                    if t != exp and len(bad) < 5:
                        bad.append({"read": k * per + j, "cpu": t[-120:].decode(errors="replace"), "gpu": exp[-120:].decode(errors="replace")})
        pool.close()
        out["parity_checked"] = checked
        out["parity_ok"] = not bad
        if bad:
            out["parity_mismatches"] = bad
            rc = 3
    if cpu is not None:
        cpu["build"] = oracle_build
        cpu["host_cpus"] = {"usable": cores, "visible": os.cpu_count(), "limit": cores_why}
    out["cpu_baseline"] = cpu
    print(json.dumps(out), flush=True)
    sys.exit(rc)


if __name__ == "__main__":
    main()
