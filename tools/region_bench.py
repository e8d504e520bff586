#!/usr/bin/env python3
"""Throughput OUTSIDE the headline's exact shape (VERDICT r4 "the headline is a point, not a region"): the same streaming
engine on other read lengths, score parameters and path counts, with the sweep kernel that ran and a parity check of a
few reads against the oracle.  One JSON line per case.

    python tools/region_bench.py [case ...]        cases: c5 x6 m3x5 hoxd70 len1500 p128 len600 (default: all)

Every case: TILES tiles of TILE reads through one rg_stream (3 handles), timed after one warm-up tile per handle."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = {
    # name: (mode, rows, paths, read length, score kwargs, tile reads, tiles)
    "c5": (8, 10000, 32, 1000, {}, 4096, 6),
    "x6": (8, 10000, 32, 1000, {"mm": (2, -6)}, 4096, 6),
    "m3x5": (8, 10000, 32, 1000, {"mm": (3, -5)}, 4096, 6),
    "hoxd70": (8, 10000, 32, 1000, {"mtx": "HOXD70.mtx"}, 2048, 4),
    "len1500": (8, 15000, 32, 1500, {}, 2048, 6),
    "len600": (8, 6000, 32, 600, {}, 4096, 6),
    "p128": (8, 10000, 128, 1000, {}, 2048, 6),
    "m4_len1500": (4, 15000, 32, 1500, {}, 2048, 6),
    "len5000": (8, 50000, 32, 5000, {}, 256, 4),
}


def main():
    from recgraph_amd import api, synth
    from oracle import oracle as O
    names = [a for a in sys.argv[1:] if not a.startswith("-")] or list(CASES)
    check = 12
    handles = int(os.environ.get("RG_REGION_HANDLES", "3"))      # (1: the kernels' lone durations)
    opts = [kv.split("=") for kv in os.environ.get("RG_REGION_OPTS", "").split(",") if kv]      # e.g. sweep_i32=1,no_retire=1
    for k, v in opts:
        api.set_option(k, int(v))
    for name in names:
        mode, rows, paths, rlen, sk, tile, tiles = CASES[name]
        g = synth.haplotype_graph(rows, paths, path_len=rlen, seed=1234)
        gfa = g.gfa()
        sm = None
        if "mm" in sk:
            sm = api.create_score_matrix_i32(*sk["mm"])
        elif "mtx" in sk:
            sm = api.create_score_matrix_i32(matrix_file_path=os.path.join(ROOT, "tests", "golden", sk["mtx"]))
        gg = api.Graph.from_gfa_text(gfa)
        params = api.make_params(mode, score_matrix=sm)
        sets = [api.Batch.pack_reads(synth.haplotype_reads(g, tile, length=rlen, seed=900 + k, mosaic_frac=0.5 if mode == 8 else 0.0)) for k in range(min(tiles, 3))]
        st = api.Stream(gg, params, device_ids=[0], handles_per_device=handles, tile_reads=tile)
        for k in range(3):
            st.push(sets[k % len(sets)])
        for k in range(3):
            st.next()
        k0 = st.kernel_stats()
        t0 = time.perf_counter()
        for k in range(tiles):
            st.push(sets[k % len(sets)])
        got = [st.next() for _ in range(tiles)]
        dt = time.perf_counter() - t0
        k1 = st.kernel_stats()
        ks = {k: round((v[0] - k0.get(k, (0, 0))[0]) / tiles, 2) for k, v in k1.items() if not k.startswith(("host:", "mem:"))}
        st.close()
        # parity: reads spread over the whole index range of the LAST timed tile (every wave slot of a full launch: the round's
        # one wrong result only showed behind index 768 of a 4096-read launch), against the oracle's threaded runner
        og = O.Graph.from_gfa_text(gfa)
        osc = None if sm is None else O.scores_from_dict({k: int(v) for k, v in sm.items()})
        omode = {4: O.M4_ABS, 8: O.M8_ABS}[mode]
        last = got[-1]
        last_reads = synth.haplotype_reads(g, tile, length=rlen, seed=900 + ((tiles - 1) % len(sets)), mosaic_frac=0.5 if mode == 8 else 0.0)
        idx = sorted(set([0, tile - 1] + [int(k * (tile - 1) / (check - 1)) for k in range(check)]))
        kw = {} if osc is None else {"scores": osc}
        _, _, exp = og.bench_text(omode, [last_reads[i] for i in idx], nthreads=min(os.cpu_count() or 1, 32), name_prefix="x", **kw)
        ok = True
        for k, i in enumerate(idx):
            e = exp[k].decode().replace("x%d\t" % k, "read%d\t" % (last.first + i), 1)
            # (bench_text numbers the reads by their position in the subset: everything but the trailing read index)
            ok = ok and last.text_of(i).decode().rsplit("\t", 1)[0] == e.rsplit("\t", 1)[0]
        sweeps = sorted(k for k in ks if k.startswith("k_sweep"))
        cu = sum(t.cell_updates for t in got)
        cp = sum(t.cell_updates_performed for t in got)
        print(json.dumps({"case": name, "mode": mode, "rows": gg.rows, "paths": paths, "read_len": rlen, "scores": sk or "default",
                          "tile_reads": tile, "tiles": tiles, "handles": handles, "options": dict((k, int(v)) for k, v in opts), "reads_per_s": round(tile * tiles / dt, 1), "ms_per_tile": round(dt / tiles * 1e3, 2),
                          "cell_updates_per_s": round(cu / dt), "performed_over_counted": round(cp / cu, 3) if cu else None, "sweep_kernels": sweeps, "kernel_ms_per_tile": ks, "parity_checked": len(idx) if ok else "FAILED"}), flush=True)


if __name__ == "__main__":
    main()
