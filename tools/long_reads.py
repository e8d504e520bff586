#!/usr/bin/env python3
"""Throughput of the striped long-read path: N reads of LEN bases against a ROWS-row / P-path synthetic
haplotype graph, -m 4 and -m 8, through one rg_batch; prints one JSON line per mode with the per-kernel times.

    python tools/long_reads.py [--reads 1024] [--len 5000] [--rows 10000] [--paths 8] [--modes 4,8] [--check 8]

--check K compares the first K records with the oracle (test infrastructure, CPU; slow at this size: keep K small).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1024)
    ap.add_argument("--len", type=int, default=5000)
    ap.add_argument("--rows", type=int, default=10000)
    ap.add_argument("--paths", type=int, default=8)
    ap.add_argument("--modes", default="4,8")
    ap.add_argument("--check", type=int, default=0)
    ap.add_argument("--repeat", type=int, default=2)
    a = ap.parse_args()
    from recgraph_amd import api, synth
    g = synth.haplotype_graph(a.rows, a.paths, path_len=a.len, seed=71)
    reads = synth.haplotype_reads(g, a.reads, length=a.len, seed=72, mosaic_frac=0.5)
    gg = api.Graph.from_gfa_text(g.gfa())
    names = ["r%d" % i for i in range(len(reads))]
    for m in [int(x) for x in a.modes.split(",")]:
        params = api.make_params(m)
        b = api.Batch(gg, reads, params)
        best = None
        for _ in range(a.repeat):
            t0 = time.perf_counter()
            b.run()
            b.fetch()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        ks = b.kernel_stats()
        line = {"mode": m, "reads": len(reads), "read_len": a.len, "graph_rows": gg.rows, "paths": a.paths,
                "seconds": round(best, 4), "reads_per_s": round(len(reads) / best, 1),
                "cell_updates_per_s": round(b.cell_updates / best, 0),
                "kernel_ms": {k: round(v[0], 2) for k, v in ks.items()} if isinstance(ks, dict) else ks}
        if a.check:
            from oracle import oracle
            og = oracle.Graph.from_gfa_text(g.gfa())
            omode = {4: oracle.M4_ABS, 5: oracle.M5_ABS, 8: oracle.M8_ABS, 9: oracle.M9_ABS}[m]
            texts = b.format_all(names).decode().splitlines(keepends=True)
            ok = all(texts[i] == og.align(omode, reads[i], name=names[i], idx=i + 1)[0] for i in range(a.check))
            line["checked"] = a.check if ok else "FAILED"
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
