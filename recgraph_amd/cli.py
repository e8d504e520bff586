"""Command line front-end mirroring the mode dispatch of the reference's ``src/main.rs`` for the modes on
the accelerated path (``-m 0, 1, 2, 3, 4, 5, 8, 9``): same positional arguments, flag names and defaults
(``src/args_parser.rs:3-147``), GAF on stdout (or written to ``-o`` with the reference's create/append rule, ``utils.rs:200-219``), ``Done in N.`` on stderr.

    python -m recgraph_amd.cli reads.fa graph.gfa -m 8 -R 4 -r 0.1 -B 1
"""
import argparse
import sys
import time


def get_sequences(path):
    """sequences::get_sequences (sequences.rs:5-45) through the library (rg_reads_from_fasta): (sequences, names)."""
    from . import api
    try:
        r = api.Reads.from_fasta(path)
    except api._lib.RecGraphError as ex:
        raise SystemExit(str(ex).split(": ", 1)[-1])     # "wrong fasta file format" (sequences.rs:41-43)
    return r.sequences(), r.names


def write_gaf_records(out_file, records, numbers):
    """utils::write_gaf (utils.rs:200-219) applied to every record in turn, with the file opened once: the reference
    appends when the file exists and ``number != 1`` and otherwise re-creates (truncates) it.  With the 0-based numbers
    of modes 4/5/8/9 the SECOND read has number 1: the file is truncated there and the first record is lost; the first
    read (number 0) is appended to a file that already exists.  Same final bytes as the per-read open/append loop."""
    import os
    exists = os.path.exists(out_file)
    mode, buf = None, []
    for rec, num in zip(records, numbers):
        if exists and num != 1:
            mode = mode or "a"
            buf.append(rec)
        else:
            mode, buf = "w", [rec]
        exists = True
    if mode:
        with open(out_file, mode) as f:
            f.write("".join(r + "\n" for r in buf))


def build_parser():
    p = argparse.ArgumentParser(prog="recgraph-hip", description="RecGraph DP hot path on MI355X")
    p.add_argument("sequence_path")
    p.add_argument("graph_path")
    p.add_argument("-o", "--out_file", default="standard output")
    p.add_argument("-m", "--aln-mode", type=int, default=0, dest="alignment_mode")
    p.add_argument("-M", "--match", type=int, default=2, dest="match_score")
    p.add_argument("-X", "--mismatch", type=int, default=4, dest="mismatch_score")
    p.add_argument("-t", "--matrix", default="none")
    p.add_argument("-O", "--gap-open", type=int, default=4)
    p.add_argument("-E", "--gap-ext", type=int, default=2, dest="gap_extension")
    p.add_argument("-r", "--multi-rec-cost", type=float, default=0.1)
    p.add_argument("-R", "--base-rec-cost", type=int, default=4)
    p.add_argument("-B", "--rec-band-width", type=float, default=1.0)
    p.add_argument("-s", "--amb-strand", default="false", choices=["true", "false"])
    p.add_argument("-b", "--extra-b", type=int, default=1)
    p.add_argument("-f", "--extra-f", type=float, default=0.01)
    p.add_argument("--scalar", action="store_true", help="-m 0 / -m 1 with the non-AVX2 path of the reference")
    p.add_argument("--devices", default="", help="comma-separated HIP device ids (default: every visible device)")
    p.add_argument("--handles", type=int, default=0, help="batch handles per device (default 3)")
    p.add_argument("--tile", type=int, default=0, help="reads per tile (default 4096 pathwise / 8192 POA)")
    p.add_argument("--queue", type=int, default=4, help="tiles that may wait in the stream's queue before the file reader waits (0: unbounded)")
    p.add_argument("--hold-mb", type=int, default=64, dest="hold_mb", help="MB of finished GAF text the stream may hold before its workers wait (0: unbounded)")
    p.add_argument("--timing", action="store_true", help="phase timings as one JSON line on stderr")
    return p


def main(argv=None):
    t0 = time.time()
    a = build_parser().parse_args(argv)
    phases = {}

    def mark(name, since):
        phases[name] = round(time.time() - since, 3)
        return time.time()
    from . import api
    tp = mark("import", t0)
    if a.alignment_mode not in (0, 1, 2, 3, 4, 5, 8, 9):
        raise SystemExit("Alignment mode must be in [0..5] or [8, 9]")   # main.rs:315-317
    amb = a.amb_strand == "true" and a.alignment_mode in (0, 1, 2, 3)     # modes 4+ ignore -s (main.rs:254-313)
    if a.matrix in ("none",):
        scores = api.create_score_matrix_i32(a.match_score, -a.mismatch_score)   # args_parser.rs:155
    else:
        scores = api.create_score_matrix_i32(matrix_file_path=a.matrix if a.matrix.endswith(".mtx") else a.matrix + ".mtx")
    g = api.Graph.from_gfa(a.graph_path)
    tp = mark("graph", tp)
    mode = {0: api.MODE_GLOBAL_POA_SCALAR if a.scalar else api.MODE_GLOBAL_POA, 2: api.MODE_GAP_POA,
            1: api.MODE_LOCAL_POA_SCALAR if a.scalar else api.MODE_LOCAL_POA, 3: api.MODE_GAP_LOCAL_POA,
            4: api.MODE_PATHWISE, 5: api.MODE_PATHWISE_SEMI, 8: api.MODE_RECOMBINATION,
            9: api.MODE_RECOMBINATION_SEMI}[a.alignment_mode]
    kw = dict(score_matrix=scores, o=-a.gap_open, e=-a.gap_extension, b=float(a.extra_b), f=a.extra_f,
              R=a.base_rec_cost, r=a.multi_rec_cost, B=a.rec_band_width)
    to_file = a.out_file != "standard output"

    def emit(first, texts):
        """texts: the stdout text of consecutive reads starting at read `first`.  With -o the records of the tile go
        through write_gaf's create / append rule right away: what precedes a read the reference panics on is in the
        file, as it would be after the reference's own per-read loop."""
        if not to_file:
            sys.stdout.write("".join(texts))
            return
        records, numbers = [], []
        # warning lines are println!'d by the exec functions whatever -o says; only the record goes through write_gaf
        for k, t in enumerate(texts):
            lines = t.split("\n")[:-1]
            sys.stdout.write("".join(ln + "\n" for ln in lines[:-1]))
            records.append(lines[-1])
            # main.rs passes i + 1 in modes 0-3 (:98-103, :161-166, :206-211, :246-251) and the 0-based i in modes
            # 4, 5, 8, 9 (:260, :268, :311)
            numbers.append(first + k + 1 if a.alignment_mode in (0, 1, 2, 3) else first + k)
        write_gaf_records(a.out_file, records, numbers)

    # The reference's read loop (main.rs:56,174,257,297) has no order dependence: the streaming engine cuts the reads into
    # tiles that the batch handles of every visible GPU (or --devices) pull from one queue; text in input order.  Like that
    # loop the process holds a bounded part of the read set: a feeder thread hands the file over block by block
    # (sequences.rs:5-45 runs inside the library), pushes wait while --queue tiles are queued, the workers wait while
    # more than --hold-mb of finished text has not been taken.  `-s true` (modes 0-3) is the stream's amb_strand option.
    import threading
    devs = [int(x) for x in a.devices.split(",")] if a.devices else None
    st = api.Stream(g, api.make_params(mode, **kw), device_ids=devs, handles_per_device=a.handles, tile_reads=a.tile,
                    amb_strand=amb, max_queued_tiles=a.queue, max_undelivered_bytes=a.hold_mb << 20)
    tp = mark("stream_create", tp)
    feed_err = []

    def feeder():
        try:
            with open(a.sequence_path, "rb") as f:
                while True:
                    block = f.read(4 << 20)
                    st.feed_fasta(block, final=not block)
                    if not block:
                        break
        except Exception as ex:                  # "wrong fasta file format" (sequences.rs:41-43), or a missing file
            feed_err.append(ex)
        finally:
            st.finish()
    th = threading.Thread(target=feeder, daemon=True)
    th.start()
    # The reference parses the whole file before it aligns the first read and panics on a malformed one (sequences.rs:41-43):
    # a scan of the file (counts only, memory speed) runs beside the first tiles, and nothing is written before it is done.
    scan = {}

    def scanner():
        try:
            scan["n"] = api.fasta_check(a.sequence_path)
        except Exception as ex:
            scan["err"] = ex
    sc = threading.Thread(target=scanner, daemon=True)
    sc.start()
    def bail(msg):
        # every early exit stops the stream FIRST: the feeder may be blocked in a bounded push and must be out of the library
        # before the stream is freed at interpreter teardown
        sys.stdout.flush()
        st.abort()
        th.join()
        st.close()
        raise SystemExit(msg)

    try:
        for t in st:
            if "first_tile" not in phases:
                mark("first_tile", tp)
                sc.join()
                if "err" in scan:
                    ex = scan["err"]
                    bail(str(ex).split(": ", 1)[-1] if isinstance(ex, api._lib.RecGraphError) else str(ex))
            bad = (t.status & (api.READ_WOULD_PANIC | api.READ_BAD_BASE)).nonzero()[0]
            n_ok = int(bad[0]) if len(bad) else t.n
            if n_ok == t.n and not to_file:
                sys.stdout.buffer.write(t.text)
            else:
                emit(t.first, [t.text_of(i).decode() for i in range(n_ok)])
            if len(bad):
                bail("read %d: the reference panics on this input" % (t.first + n_ok))
    except api._lib.RecGraphError as ex:         # a failed tile
        bail(str(ex))
    th.join()
    sc.join()
    if "err" in scan:        # (a file without a single complete read: no tile ever came)
        feed_err.insert(0, scan["err"])
    if feed_err:
        ex = feed_err[0]
        raise SystemExit(str(ex).split(": ", 1)[-1] if isinstance(ex, api._lib.RecGraphError) else str(ex))
    tp = mark("all_tiles", tp)
    phases["kernels_ms"] = {k: round(v[0], 1) for k, v in st.kernel_stats().items()}
    st.close()
    tp = mark("stream_close", tp)
    sys.stdout.flush()
    if a.timing:
        import json
        phases["total"] = round(time.time() - t0, 3)
        sys.stderr.write(json.dumps(phases) + "\n")
    sys.stderr.write("Done in %d.\n" % int(time.time() - t0))    # main.rs:319-323


if __name__ == "__main__":
    main()
