"""-m 4 / -m 8 parity on the GPU: HIP pipeline (through the C ABI) vs the oracle, byte-for-byte."""
import numpy as np
import os

import pytest

pytestmark = pytest.mark.gpu
SLOW = os.environ.get("RG_SLOW_TESTS", "0") not in ("", "0")      # the long variants of the trimmed tests


SPEC_MARGIN_DEFAULT = 112     # rg_host.hpp (what the tests put back after changing it)

def _switch(var, on):
    """The diagnostic switches of the library (rg_set_option; the environment variables of the same names only set the
    defaults when the library is loaded)."""
    from recgraph_amd import api
    api.set_option({"RG_SWEEP_I32": "sweep_i32", "RG_NO_FREC": "no_frec", "RG_THREE_SWEEPS": "three_sweeps", "RG_LAYER_I32": "layer_i32"}[var], on)



DIAMOND = ("H\tVN:Z:1.0\nS\t1\tA\nS\t2\tT\nS\t3\tC\nS\t4\tG\nL\t1\t+\t2\t+\t0M\nL\t1\t+\t3\t+\t0M\nL\t2\t+\t4\t+\t0M\n"
           "L\t3\t+\t4\t+\t0M\nP\tp0\t1+,2+,4+\t*\nP\tp1\t1+,3+,4+\t*\n")
TWO_BUBBLES = ("S\t1\tA\nS\t2\tT\nS\t3\tC\nS\t4\tG\nS\t5\tA\nS\t6\tC\nS\t7\tT\n" +
               "".join(f"L\t{a}\t+\t{b}\t+\t0M\n" for a, b in [(1, 2), (1, 3), (2, 4), (3, 4), (4, 5), (4, 6), (5, 7), (6, 7)]) +
               "P\tp0\t1+,2+,4+,5+,7+\t*\nP\tp1\t1+,3+,4+,6+,7+\t*\n")


def _check(oracle, gfa, reads, mode, omode, names=None, **kw):
    from recgraph_amd import api
    og = oracle.Graph.from_gfa_text(gfa)
    g = api.Graph.from_gfa_text(gfa)
    names = names or ["r%d" % i for i in range(len(reads))]
    texts, status = api.align_batch(g, reads, names, mode=mode, **kw)
    okw = {k: v for k, v in kw.items() if k in ("R", "r", "B")}
    bad = []
    for i, rd in enumerate(reads):
        exp = og.align(omode, rd, name=names[i], idx=i + 1, **okw)[0]
        if texts[i] != exp:
            bad.append((i, rd[:40], texts[i][-300:], exp[-300:]))
    assert not bad, (len(bad), bad[:2])
    return texts


def test_hand_derived_vectors(oracle):
    from recgraph_amd import api
    t = _check(oracle, DIAMOND, ["ATG"], api.MODE_PATHWISE, oracle.M4, names=["name"])
    assert t[0] == "name\t3\t0\t2\t+\t>1>2>4\t3\t0\t2\t0\t*\t*\t3M, best path: 0, score: 6\tATG\n"
    t = _check(oracle, TWO_BUBBLES, ["ATGCT"], api.MODE_PATHWISE, oracle.M4, names=["name"])
    assert t[0] == "name\t5\t0\t4\t+\t>1>3>4>6>7\t5\t0\t4\t0\t*\t*\t1M1X3M, best path: 1, score: 4\tACGCT\n"
    t = _check(oracle, TWO_BUBBLES, ["ATGCT"], api.MODE_RECOMBINATION, oracle.M8, names=["name"])
    assert t[0] == ("name\t5\t0\t4\t+\t>1>2>4>6>7\t5\t0\t4\t0\t*\t*\t5M, recombination path 0 1, nodes 2[0] 4[0], "
                    "score: 5.8, displacement: 2\tATGCT\t1\n")


def test_small_graphs_vs_literal(oracle):
    """Tiny graphs against the LITERAL (delta-encoded, unpruned O(L^2 n) search) restatement."""
    from recgraph_amd import api
    rng = np.random.default_rng(3)
    for gfa in (DIAMOND, TWO_BUBBLES):
        reads = ["".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=int(rng.integers(1, 9)))) for _ in range(40)]
        reads += ["A", "ATG", "ACG", "ATGAT", "ACGCT", "ATGCT", "TTTTTTTT", "N", "ANG"]
        _check(oracle, gfa, reads, api.MODE_PATHWISE, oracle.M4)
        _check(oracle, gfa, reads, api.MODE_RECOMBINATION, oracle.M8)
        _check(oracle, gfa, reads, api.MODE_RECOMBINATION, oracle.M8, R=0, r=0.0)
        _check(oracle, gfa, reads, api.MODE_RECOMBINATION, oracle.M8, R=1, r=0.5, B=0.6)


def test_example_data(oracle, example_gfa, example_reads):
    from recgraph_amd import api
    names, reads = example_reads
    _check(oracle, example_gfa, reads, api.MODE_PATHWISE, oracle.M4_ABS, names=names)
    _check(oracle, example_gfa, reads, api.MODE_RECOMBINATION, oracle.M8_ABS, names=names)
    _check(oracle, example_gfa, reads[:4], api.MODE_RECOMBINATION, oracle.M8_PRUNED, names=names[:4])


def test_synthetic_haplotype_graph(oracle):
    from recgraph_amd import api, synth
    g = synth.haplotype_graph(1500, 8, path_len=300, seed=11)
    reads = synth.haplotype_reads(g, 48, length=300, seed=12, mosaic_frac=0.5)
    reads += [g.path_sequence(0)[:300], g.path_sequence(3)[:150] + g.path_sequence(5)[150:300], "ACGT" * 10]
    _check(oracle, g.gfa(), reads, api.MODE_PATHWISE, oracle.M4_ABS)
    _check(oracle, g.gfa(), reads, api.MODE_RECOMBINATION, oracle.M8_ABS)
    _check(oracle, g.gfa(), reads[:6], api.MODE_RECOMBINATION, oracle.M8_PRUNED)


def test_semiglobal_modes(oracle, example_gfa, example_reads):
    """-m 5 (pathwise_alignment_semiglobal.rs) and -m 9 (aln_mode 9 branches)."""
    from recgraph_amd import api, synth
    rng = np.random.default_rng(4)
    for gfa in (DIAMOND, TWO_BUBBLES):
        reads = ["".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=int(rng.integers(1, 9)))) for _ in range(40)]
        reads += ["A", "T", "ATG", "TG", "GCT", "ATGCT", "TTTTTTTT", "N"]
        _check(oracle, gfa, reads, api.MODE_PATHWISE_SEMI, oracle.M5)
        _check(oracle, gfa, reads, api.MODE_RECOMBINATION_SEMI, oracle.M9)
        _check(oracle, gfa, reads, api.MODE_RECOMBINATION_SEMI, oracle.M9, R=0, r=0.0)
        _check(oracle, gfa, reads, api.MODE_RECOMBINATION_SEMI, oracle.M9, R=1, r=0.5, B=0.6)
    names, reads = example_reads
    _check(oracle, example_gfa, reads, api.MODE_PATHWISE_SEMI, oracle.M5_ABS, names=names)
    _check(oracle, example_gfa, reads, api.MODE_RECOMBINATION_SEMI, oracle.M9_ABS, names=names)
    _check(oracle, example_gfa, reads[:3], api.MODE_RECOMBINATION_SEMI, oracle.M9_PRUNED, names=names[:3])
    g = synth.haplotype_graph(1500, 8, path_len=300, seed=11)
    rd = synth.haplotype_reads(g, 32, length=120, seed=13, mosaic_frac=0.5)
    rd += [g.path_sequence(2)[50:170], g.path_sequence(3)[100:160] + g.path_sequence(5)[160:230], "ACGT" * 10]
    _check(oracle, g.gfa(), rd, api.MODE_PATHWISE_SEMI, oracle.M5_ABS)
    _check(oracle, g.gfa(), rd, api.MODE_RECOMBINATION_SEMI, oracle.M9_ABS)
    _check(oracle, g.gfa(), rd, api.MODE_RECOMBINATION_SEMI, oracle.M9_ABS, R=0, r=0.0)


def test_wide_and_long_shapes(oracle):
    """Kernel template coverage: C = 32 (reads up to 2047 bases), more than 32 paths (64-bit masks),
    general (non-uniform) gap scores, and the read-length limit as a status code."""
    from recgraph_amd import _lib, api, synth
    # long reads: C = 32 columns per lane
    g = synth.haplotype_graph(4000, 6, path_len=1400, seed=21)
    rd = synth.haplotype_reads(g, 6, length=1400, seed=22, mosaic_frac=0.5) + [g.path_sequence(1)[:1100]]
    _check(oracle, g.gfa(), rd, api.MODE_PATHWISE, oracle.M4_ABS)
    _check(oracle, g.gfa(), rd, api.MODE_RECOMBINATION, oracle.M8_ABS)
    _check(oracle, g.gfa(), rd[:3], api.MODE_RECOMBINATION_SEMI, oracle.M9_ABS)
    # 40 paths
    g = synth.haplotype_graph(1200, 40, path_len=150, seed=23)
    rd = synth.haplotype_reads(g, 24, length=150, seed=24, mosaic_frac=0.5)
    _check(oracle, g.gfa(), rd, api.MODE_PATHWISE, oracle.M4_ABS)
    _check(oracle, g.gfa(), rd, api.MODE_RECOMBINATION, oracle.M8_ABS)
    _check(oracle, g.gfa(), rd[:6], api.MODE_RECOMBINATION, oracle.M8_PRUNED)
    # non-uniform gap costs (a custom matrix as api.rs callers may pass): the general-GP kernel variant
    sm = api.create_score_matrix_i32(3, -5)
    sm[("A", "-")] = -7; sm[("-", "A")] = -7; sm[("G", "-")] = -12; sm[("-", "G")] = -12
    og = oracle.Graph.from_gfa_text(g.gfa())
    gg = api.Graph.from_gfa_text(g.gfa())
    table = api._table_from_dict(sm)
    for mode, om in ((api.MODE_PATHWISE, oracle.M4_ABS), (api.MODE_RECOMBINATION, oracle.M8_ABS)):
        texts, _ = api.align_batch(gg, rd[:10], None, mode=mode, score_matrix=sm)
        for i, r in enumerate(rd[:10]):
            assert texts[i] == og.align(om, r, name="read%d" % i, scores=table)[0]
    # reads longer than the supported 16383 bases: status code, no abort
    with pytest.raises(_lib.RecGraphError):
        api.align_batch(gg, ["ACGT" * 4200], None, mode=api.MODE_PATHWISE)


def test_sweep_kernel_variants_agree(oracle):
    """The packed 16-bit sweep (default when the scores fit), the i32 sweep (RG_SWEEP_I32) and the Cand-list forward
    emission (RG_NO_FREC) are three implementations of the same DP: byte-identical records, all equal to the oracle.
    A matrix whose scores do not fit 16 bits must take the i32 kernel by itself."""
    from recgraph_amd import api, synth
    g = synth.haplotype_graph(2500, 12, path_len=400, seed=31)
    rd = synth.haplotype_reads(g, 40, length=400, seed=32, mosaic_frac=0.6) + [g.path_sequence(4)[:333], "ACGTTGCA" * 9]
    gg = api.Graph.from_gfa_text(g.gfa())
    for mode, om, cut in ((api.MODE_RECOMBINATION, oracle.M8_ABS, 42), (api.MODE_RECOMBINATION_SEMI, oracle.M9_ABS, 12),
                          (api.MODE_PATHWISE, oracle.M4_ABS, 42), (api.MODE_PATHWISE_SEMI, oracle.M5_ABS, 12)):
        reads = [r[:180] for r in rd[:cut]] if mode in (api.MODE_RECOMBINATION_SEMI, api.MODE_PATHWISE_SEMI) else rd[:cut]
        base = _check(oracle, g.gfa(), reads, mode, om)
        for var in ("RG_SWEEP_I32", "RG_NO_FREC", "RG_LAYER_I32"):
            _switch(var, 1)
            texts, _ = api.align_batch(gg, reads, ["r%d" % i for i in range(len(reads))], mode=mode)
            _switch(var, 0)
            assert texts == base, var
    # a narrow recombination band and very short reads (thresholds of "never" columns, rows every path visits)
    tiny = ["GG", "A", "ACG", "TTTTT", rd[0][:7], rd[1][:30]] + rd[:6]
    for mode, om in ((api.MODE_RECOMBINATION, oracle.M8_ABS), (api.MODE_RECOMBINATION_SEMI, oracle.M9_ABS)):
        base = _check(oracle, g.gfa(), tiny, mode, om, R=0, r=0.1, B=0.8)
        for var in ("RG_SWEEP_I32", "RG_NO_FREC", "RG_THREE_SWEEPS"):
            _switch(var, 1)
            texts, _ = api.align_batch(gg, tiny, ["r%d" % i for i in range(len(tiny))], mode=mode, R=0, r=0.1, B=0.8)
            _switch(var, 0)
            assert texts == base, (var, mode)
    # scores outside the 16-bit budget: (rows on a path + read length) * max |score| > 24000
    sm = api.create_score_matrix_i32(90, -120)
    table = api._table_from_dict(sm)
    og = oracle.Graph.from_gfa_text(g.gfa())
    texts, _ = api.align_batch(gg, rd[:8], None, mode=api.MODE_RECOMBINATION, score_matrix=sm)
    for i, r in enumerate(rd[:8]):
        assert texts[i] == og.align(oracle.M8_ABS, r, name="read%d" % i, scores=table)[0]


def _sweep16_admissible(scores36, max_path_rows, max_n, C):
    """The admission rule of rg_sweep16.hip (sweep16_admissible), restated: the range of what the rows store."""
    t = scores36
    if any(t[b * 6 + 5] != t[5] for b in range(1, 5)) or any(t[b * 6 + 5] > 0 or t[5 * 6 + b] > 0 for b in range(5)):
        return False
    ent = [t[x * 6 + y] for x in range(6) for y in range(6) if not (x == 5 and y == 5)]
    sub = [t[x * 6 + y] for x in range(5) for y in range(5)]
    maxabs = max(abs(v) for v in ent)
    if maxabs > 1000:
        return False
    g, rows, n = t[5], max_path_rows + 2, max_n + 2
    zlo = rows * g - n * max(0, g - min(sub))
    zhi = n * max(0, max(sub) - g)
    return zlo >= -29000 and zhi <= 29000 and zhi - zlo <= 32000 and (rows + n) * maxabs <= 32000 and (C // 2 + 2) * maxabs <= 2000


def test_scores_at_the_edge_of_the_16_bit_budget(oracle):
    """`sweep16_admissible` since round 5 bounds what the rows STORE (z = A - c g), not A: -X 6 / -X 7 and -M 3 -X 5 at 1 kbp run
    packed, -X 8 does not; shorter reads admit larger scores.  Every case prints the oracle's bytes on whichever side of the line
    it falls, and the kernel statistics say which sweep ran — the one the rule (restated above) predicts."""
    from recgraph_amd import api, synth
    cases = ((9000, 12, 1000, (2, -6)), (9000, 12, 1000, (3, -5)), (9000, 12, 1000, (2, -7)), (9000, 12, 1000, (2, -8)),
             (4500, 8, 500, (2, -14)), (4500, 8, 500, (2, -16)), (4500, 8, 500, (5, -3)))
    seen = set()
    for rows, P, plen, (m, x) in cases:
        g = synth.haplotype_graph(rows, P, path_len=plen, seed=400 + plen + abs(x))
        rd = synth.haplotype_reads(g, 7, length=plen, seed=401 + abs(x), mosaic_frac=0.6) + [g.path_sequence(P - 1)[:plen]]
        sm = api.create_score_matrix_i32(m, x)
        osc = oracle.scores_from_dict({k: int(v) for k, v in sm.items()})
        gg = api.Graph.from_gfa_text(g.gfa())
        og = oracle.Graph.from_gfa_text(g.gfa())
        path_rows = max(len(lst.split(",")) for lst in gg.dump(30).strip(";").split(";") if lst)
        n = max(len(r) for r in rd)
        C = 4
        while C * 64 < n + 1 and C < 32:
            C *= 2
        packed = _sweep16_admissible(osc, path_rows, n, C)
        seen.add(packed)
        names = ["q%d" % i for i in range(len(rd))]
        for mode, om in ((api.MODE_RECOMBINATION, oracle.M8_ABS), (api.MODE_PATHWISE, oracle.M4_ABS)):
            b = api.Batch(gg, rd, api.make_params(mode, score_matrix=sm))
            b.run()
            b.fetch()
            ks = b.kernel_stats()
            assert any(k.startswith("k_sweep16") for k in ks) == packed and any(k.startswith("k_sweep_") for k in ks) == (not packed), (m, x, plen, path_rows, sorted(ks))
            for i, r in enumerate(rd):
                exp = og.align(om, r, name=names[i], idx=i + 1, scores=osc)[0]
                assert b.gaf_text(i, names[i], i + 1) == exp, (m, x, plen, mode, i)
    assert seen == {True, False}
    assert _sweep16_admissible(oracle.scores_match_mis(2, -6), 1000, 1000, 16) and not _sweep16_admissible(oracle.scores_match_mis(2, -8), 1000, 1000, 16)


def test_packed_opt0_equals_the_i32_form(oracle):
    """`k_opt0_16` (the speculative / provable forward bound on packed rows) against `k_opt0` on every read: the bound only
    steers the pruning — a wrong one would cost speed or a second pass, never bytes — so the driver's RG_DEBUG mode runs both
    and fails the batch on any difference.  Global and semiglobal, one- and two-path picks, the provable path-0 bound, reads
    of 16 / 32 columns per lane, non-default scores."""
    from recgraph_amd import api, synth
    cases = ((2500, 12, 400, 71, None), (4000, 32, 700, 73, None), (3000, 8, 1300, 74, None), (2500, 12, 400, 75, api.create_score_matrix_i32(3, -5)))
    try:
        api.set_option("debug", 1)
        for rows, P, plen, seed, sm in cases:
            g = synth.haplotype_graph(rows, P, path_len=plen, seed=seed)
            rd = synth.haplotype_reads(g, 24, length=plen, seed=seed + 1, mosaic_frac=0.6) + [g.path_sequence(P - 1)[:plen // 2], "ACGT" * 5, "T"]
            gg = api.Graph.from_gfa_text(g.gfa())
            names = ["r%d" % i for i in range(len(rd))]
            kw = {} if sm is None else {"score_matrix": sm}
            for mode in (api.MODE_RECOMBINATION, api.MODE_RECOMBINATION_SEMI):
                for opts in ((), (("no_pick2", 1),), (("no_spec", 1),)):
                    try:
                        for name, val in opts:
                            api.set_option(name, val)
                        texts, status = api.align_batch(gg, rd, names, mode=mode, **kw)       # raises if the two kernels disagree
                    finally:
                        for name, _ in opts:
                            api.set_option(name, 0)
                    assert len(texts) == len(rd)
    finally:
        api.set_option("debug", 0)


def test_speculative_forward_bound(oracle):
    """-m 8, two-sweep record pipeline, P <= 64: the forward sweep prunes with a SPECULATIVE bound (score against the path
    k_pick votes for, minus a margin) that k_verify checks afterwards; reads whose search maximum stayed below it are
    aligned again with the provable bound.  Same bytes with the speculation off (no_spec), with the default margin, with
    a zero margin, and with a margin that makes EVERY read fail the check (all of them go through the second pass)."""
    from recgraph_amd import api, synth
    g = synth.haplotype_graph(2500, 12, path_len=400, seed=71)
    rd = synth.haplotype_reads(g, 60, length=400, seed=72, mosaic_frac=0.6) + [g.path_sequence(4)[:333], "ACGTTGCA" * 9, "A", "GG"]
    gg = api.Graph.from_gfa_text(g.gfa())
    names = ["r%d" % i for i in range(len(rd))]
    base = _check(oracle, g.gfa(), rd, api.MODE_RECOMBINATION, oracle.M8_ABS)
    try:
        for margin in (0, 40, -1000000):
            api.set_option("spec_margin", margin)
            texts, _ = api.align_batch(gg, rd, names, mode=api.MODE_RECOMBINATION)
            assert texts == base, margin
            for R, r, B in ((0, 0.1, 0.8), (9, 0.5, 1.0)):
                exp = _check(oracle, g.gfa(), rd[:20], api.MODE_RECOMBINATION, oracle.M8_ABS, R=R, r=r, B=B)
                assert exp is not None
        api.set_option("spec_margin", SPEC_MARGIN_DEFAULT)
        api.set_option("no_spec", 1)
        texts, _ = api.align_batch(gg, rd, names, mode=api.MODE_RECOMBINATION)
        assert texts == base
    finally:
        api.set_option("spec_margin", SPEC_MARGIN_DEFAULT)
        api.set_option("no_spec", 0)
    # a stream tile with forced failures in two chunks (chunk_reads) keeps the input order
    try:
        api.set_option("spec_margin", -1000000)
        api.set_option("chunk_reads", 24)
        texts, _ = api.align_stream(gg, rd, names, mode=api.MODE_RECOMBINATION, device_ids=[0], handles_per_device=2, tile_reads=40)
        assert texts == base
    finally:
        api.set_option("spec_margin", SPEC_MARGIN_DEFAULT)
        api.set_option("chunk_reads", 0)


def test_scores_beyond_the_32_bit_keys_are_refused():
    """the i32 sweep packs (value, path) keys as value * 256 + path: a batch whose scores could reach 2^23 in
    magnitude is refused with RG_ERR_CAPACITY (-5) instead of wrapping silently; just below the limit it still runs."""
    from recgraph_amd import _lib, api, synth
    g = synth.haplotype_graph(1200, 4, path_len=400, seed=91)
    rd = synth.haplotype_reads(g, 4, length=400, seed=92, mosaic_frac=0.5)
    gg = api.Graph.from_gfa_text(g.gfa())
    with pytest.raises(_lib.RecGraphError) as e:
        api.align_batch(gg, rd, None, mode=api.MODE_RECOMBINATION, score_matrix=api.create_score_matrix_i32(6000, -6000))
    assert e.value.code == -5 and "2^23" in str(e.value)
    texts, status = api.align_batch(gg, rd, None, mode=api.MODE_PATHWISE, score_matrix=api.create_score_matrix_i32(2000, -2500))
    assert len(texts) == 4 and not any(status)


def test_gather_runs(oracle):
    """k_sweep16 gather runs: long runs of inner rows that a wide group (>= 8 paths) goes through are processed as the alpha
    + a column map, the members once per run.  Graphs with long segments shared by many paths, global and semiglobal, with
    and without the switch (`no_gather`): byte-identical records, equal to the oracle."""
    from recgraph_amd import api, synth
    for P, rows, plen, shared, seed in ((12, 1400, 300, 0.8, 101), (32, 2600, 400, 0.6, 102), (60, 1800, 260, 0.9, 103)):
        g = synth.haplotype_graph(rows, P, path_len=plen, seed=seed, shared_frac=shared)
        rd = synth.haplotype_reads(g, 24, length=plen, seed=seed + 1, mosaic_frac=0.6) + [g.path_sequence(P - 1)[:plen - 5], "ACGT" * 6]
        gg = api.Graph.from_gfa_text(g.gfa())
        names = ["r%d" % i for i in range(len(rd))]
        for mode, om, reads in ((api.MODE_RECOMBINATION, oracle.M8_ABS, rd), (api.MODE_RECOMBINATION_SEMI, oracle.M9_ABS, [r[:plen * 2 // 3] for r in rd[:10]])):
            base = _check(oracle, g.gfa(), reads, mode, om)
            try:
                api.set_option("no_gather", 1)
                texts, _ = api.align_batch(gg, reads, names[:len(reads)], mode=mode)
            finally:
                api.set_option("no_gather", 0)
            assert texts == base, (P, mode)
        _check(oracle, g.gfa(), rd[:6], api.MODE_RECOMBINATION, oracle.M8_ABS, R=0, r=0.1, B=0.8)


def test_split_step_tables(oracle):
    """k_sweep16 split step tables: a group of a several-group row whose paths are exactly the paths of a register run on its
    predecessor row is processed as the TAIL of that run (rows in registers), the row's keys folding across its groups.
    Allele blocks between shared segments (the shape of configs 4 / 5) with few and many alleles, with and without the
    switch (`no_split`): byte-identical records, equal to the oracle; with the speculative bound off too (loose forward
    thresholds: every row emits)."""
    from recgraph_amd import api, synth
    for P, rows, plen, shared, seed in ((32, 3000, 300, 0.3, 111), (16, 900, 250, 0.3, 112), (6, 700, 220, 0.4, 113), (40, 1500, 200, 0.15, 114)):
        g = synth.haplotype_graph(rows, P, path_len=plen, seed=seed, shared_frac=shared)
        rd = synth.haplotype_reads(g, 20, length=plen, seed=seed + 1, mosaic_frac=0.6) + [g.path_sequence(P - 1)[:plen - 5], "ACGT" * 6]
        gg = api.Graph.from_gfa_text(g.gfa())
        names = ["r%d" % i for i in range(len(rd))]
        for mode, om in ((api.MODE_RECOMBINATION, oracle.M8_ABS), (api.MODE_PATHWISE, oracle.M4_ABS)):
            base = _check(oracle, g.gfa(), rd, mode, om)
            for opts in (("no_split",), ("no_spec",), ("no_split", "no_spec")):
                try:
                    for o in opts:
                        api.set_option(o, 1)
                    texts, _ = api.align_batch(gg, rd, names, mode=mode)
                finally:
                    for o in opts:
                        api.set_option(o, 0)
                assert texts == base, (P, mode, opts)
        _check(oracle, g.gfa(), rd[:6], api.MODE_RECOMBINATION, oracle.M8_ABS, R=0, r=0.1, B=0.8)


def test_random_dag_graphs(oracle):
    """Graphs that are NOT blocks of alleles between shared segments: every path a random walk through segments in id order
    (nested / overlapping bubbles, one-row segments, groups that are proper subsets of a row's paths, paths that part ways
    after a shared segment).  Every pathwise mode against the oracle, and the step-table switches (`no_split`,
    `no_gather`) and the i32 sweep against the default: byte-identical."""
    from recgraph_amd import api, synth
    for nseg, P, seed, kw in ((70, 3, 201, {}), (120, 9, 202, {"max_jump": 3}), (160, 24, 203, {"max_seg": 6}), (90, 40, 204, {"max_jump": 6, "max_seg": 14}),
                              (200, 6, 205, {"max_jump": 2, "max_seg": 3}), (60, 64, 206, {"similar": 0.8})):
        g = synth.random_dag_graph(nseg, P, seed=seed, **kw)
        plen = min(len(g.path_sequence(k)) for k in range(P))
        rd = synth.haplotype_reads(g, 10, length=plen, seed=seed + 1, mosaic_frac=0.6) + [g.path_sequence(P - 1), g.path_sequence(0)[:plen // 2], "ACGT" * 5]
        gg = api.Graph.from_gfa_text(g.gfa())
        names = ["r%d" % i for i in range(len(rd))]
        for mode, om in ((api.MODE_PATHWISE, oracle.M4_ABS), (api.MODE_RECOMBINATION, oracle.M8_ABS),
                         (api.MODE_PATHWISE_SEMI, oracle.M5_ABS), (api.MODE_RECOMBINATION_SEMI, oracle.M9_ABS)):
            base = _check(oracle, g.gfa(), rd, mode, om)
            for opt in ("no_split", "no_gather", "sweep_i32"):
                try:
                    api.set_option(opt, 1)
                    texts, _ = api.align_batch(gg, rd, names, mode=mode)
                finally:
                    api.set_option(opt, 0)
                assert texts == base, (nseg, P, mode, opt)
        _check(oracle, g.gfa(), rd[:5], api.MODE_RECOMBINATION, oracle.M8_ABS, R=0, r=0.1, B=0.8)


def test_random_dag_graphs_in_every_switch_family(oracle):
    """the round-3 split-table bug produced wrong scores on random-walk graphs only, while every test family
    that exercises a kernel switch ran on block-shaped haplotype graphs.  Here random-walk graphs (nested / overlapping
    bubbles, one-row segments, proper-subset groups) go through EVERY switch of the pathwise pipeline — i32 sweep, Cand-list
    emission, three sweeps, speculation off / zero margin / forced second pass, no gather, no split, small chunks, path
    retirement off / in one sweep only / evaluated every 16 or 4 records (so that graphs of this size retire paths at all),
    one-path picks only, and pairs of them — in all four pathwise modes: byte-identical to the default, which equals the oracle."""
    from recgraph_amd import api, synth
    cases = ((150, 12, 301, {"max_jump": 3, "max_seg": 8}), (110, 32, 302, {"max_jump": 5, "max_seg": 12, "similar": 0.7}),
             (240, 5, 303, {"max_jump": 2, "max_seg": 4}))
    switches = (("sweep_i32", 1), ("no_frec", 1), ("three_sweeps", 1), ("no_spec", 1), ("spec_margin", 0), ("spec_margin", -1000000),
                ("no_gather", 1), ("no_split", 1), ("chunk_reads", 5), ("layer_i32", 1), ("no_retire", 1), ("no_retire", 2), ("no_retire", 3),
                ("no_pick2", 1), ("no_order", 1), ("retire_shift", 4), ("retire_shift", 2),
                # direction words on demand (round 6): off; without the always-stored edge rows (1 / 1000 of the rows: reads whose
                # final paths are not the picked ones come back for the second pass); with half of the rows always stored
                ("no_dsel", 1), ("dsel_edge", 1000), ("dsel_edge", 2))
    pairs = ((("three_sweeps", 1), ("sweep_i32", 1)), (("no_split", 1), ("no_gather", 1)), (("spec_margin", -1000000), ("chunk_reads", 4)),
             (("no_frec", 1), ("no_spec", 1)), (("no_retire", 1), ("no_split", 1)), (("no_spec", 1), ("no_gather", 1)),
             (("no_pick2", 1), ("spec_margin", 0)), (("no_retire", 3), ("no_split", 1)),
             (("dsel_edge", 1000), ("no_pick2", 1)), (("dsel_edge", 1000), ("retire_shift", 4)), (("dsel_edge", 1000), ("spec_margin", 0)), (("dsel_edge", 1000), ("chunk_reads", 5)),
             # evaluation every 16 / 4 records: graphs of this size only retire paths with a short period
             (("retire_shift", 4), ("spec_margin", 0)), (("retire_shift", 3), ("no_split", 1)), (("retire_shift", 4), ("no_retire", 2)),
             (("retire_shift", 4), ("no_retire", 3)), (("retire_shift", 4), ("no_gather", 1)), (("retire_shift", 4), ("no_pick2", 1)),
             # the i32 sweep's own path retirement and speculative bound (round 5)
             (("sweep_i32", 1), ("retire_shift", 4)), (("sweep_i32", 1), ("retire_shift", 3), ("spec_margin", 0)), (("sweep_i32", 1), ("no_retire", 1)),
             (("sweep_i32", 1), ("retire_shift", 4), ("no_retire", 2)), (("sweep_i32", 1), ("retire_shift", 4), ("no_retire", 3)),
             (("sweep_i32", 1), ("retire_shift", 4), ("no_spec", 1)), (("sweep_i32", 1), ("retire_shift", 2), ("no_order", 1)))
    defaults = {"spec_margin": SPEC_MARGIN_DEFAULT, "retire_shift": 8, "dsel_edge": 8}
    for nseg, P, seed, kw in cases:
        g = synth.random_dag_graph(nseg, P, seed=seed, **kw)
        plen = min(len(g.path_sequence(k)) for k in range(P))
        rd = synth.haplotype_reads(g, 12, length=plen, seed=seed + 1, mosaic_frac=0.7) + [g.path_sequence(P - 1), g.path_sequence(1)[:plen * 2 // 3], "ACGT" * 4]
        gg = api.Graph.from_gfa_text(g.gfa())
        names = ["r%d" % i for i in range(len(rd))]
        for mode, om in ((api.MODE_RECOMBINATION, oracle.M8_ABS), (api.MODE_PATHWISE, oracle.M4_ABS),
                         (api.MODE_RECOMBINATION_SEMI, oracle.M9_ABS), (api.MODE_PATHWISE_SEMI, oracle.M5_ABS)):
            base = _check(oracle, g.gfa(), rd, mode, om)
            for combo in [(sw,) for sw in switches] + list(pairs):
                try:
                    for name, val in combo:
                        api.set_option(name, val)
                    texts, _ = api.align_batch(gg, rd, names, mode=mode)
                finally:
                    for name, _ in combo:
                        api.set_option(name, defaults.get(name, 0))
                assert texts == base, (nseg, P, mode, combo)
        for R, r, B in ((0, 0.1, 0.8), (9, 0.5, 1.0), (1, 0.0, 0.5)):
            _check(oracle, g.gfa(), rd[:8], api.MODE_RECOMBINATION, oracle.M8_ABS, R=R, r=r, B=B)


def test_kilobase_reads_on_nested_bubbles(oracle):
    """Reads of 1 kbp and more (the 16-columns-per-lane kernels of the headline configuration) on random-walk graphs with
    32 and 60 paths — the shape the block-built config-5 graph does not have — against the absolute-form oracle, default
    pipeline, a zero speculation margin, and the plain step tables."""
    from recgraph_amd import api, synth
    for nseg, P, seed, kw in ((620, 32, 311, {"max_jump": 4, "max_seg": 10}), (560, 60, 312, {"max_jump": 3, "max_seg": 9, "similar": 0.6})):
        g = synth.random_dag_graph(nseg, P, seed=seed, **kw)
        plen = min(len(g.path_sequence(k)) for k in range(P))
        assert 1000 <= plen <= 2047, plen
        rd = synth.haplotype_reads(g, 5, length=plen, seed=seed + 1, mosaic_frac=0.8) + [g.path_sequence(P - 1)[:plen]]
        gg = api.Graph.from_gfa_text(g.gfa())
        names = ["r%d" % i for i in range(len(rd))]
        base = _check(oracle, g.gfa(), rd, api.MODE_RECOMBINATION, oracle.M8_ABS)
        _check(oracle, g.gfa(), rd[:3], api.MODE_PATHWISE, oracle.M4_ABS)
        # (32 columns per lane when the reads exceed 1023 bases: the packed record variant with register runs of two rows and
        # path retirement, and — sweep_i32 — the i32 sweep's one-wave form with its own retirement; every 16 records here)
        defaults = {"spec_margin": SPEC_MARGIN_DEFAULT, "retire_shift": 8}
        for combo in ((("spec_margin", 0),), (("no_split", 1),), (("no_gather", 1),), (("retire_shift", 4),), (("retire_shift", 4), ("spec_margin", 0)),
                      (("sweep_i32", 1),), (("sweep_i32", 1), ("retire_shift", 4)), (("no_retire", 1),)):
            try:
                for name, val in combo:
                    api.set_option(name, val)
                texts, _ = api.align_batch(gg, rd, names, mode=api.MODE_RECOMBINATION)
            finally:
                for name, _ in combo:
                    api.set_option(name, defaults.get(name, 0))
            assert texts == base, (P, combo)


def test_three_sweep_pipeline(oracle):
    """The -m 8 / -m 9 pipeline the driver takes when a gap entry is positive (no path-0 lower bound for the forward
    thresholds: forward column maxima first, reverse sweep, forward again) or on request (RG_THREE_SWEEPS), with the
    packed 16-bit sweep and with the i32 sweep.  Same records as the two-sweep pipeline, all equal to the oracle."""
    from recgraph_amd import api, synth
    g = synth.haplotype_graph(2500, 12, path_len=400, seed=41)
    rd = synth.haplotype_reads(g, 30, length=400, seed=42, mosaic_frac=0.6) + [g.path_sequence(7)[:390], "ACGTTGCA" * 11]
    gg = api.Graph.from_gfa_text(g.gfa())
    names = ["r%d" % i for i in range(len(rd))]
    for mode, om, reads in ((api.MODE_RECOMBINATION, oracle.M8_ABS, rd), (api.MODE_RECOMBINATION_SEMI, oracle.M9_ABS, [r[:170] for r in rd[:14]])):
        base = _check(oracle, g.gfa(), reads, mode, om)
        _switch("RG_THREE_SWEEPS", 1)
        t16, _ = api.align_batch(gg, reads, names[:len(reads)], mode=mode)
        _switch("RG_SWEEP_I32", 1)
        t32, _ = api.align_batch(gg, reads, names[:len(reads)], mode=mode)
        _switch("RG_SWEEP_I32", 0)
        _switch("RG_THREE_SWEEPS", 0)
        assert t16 == base and t32 == base
    # matrices with positive gap entries (reachable through api.rs-style custom matrices only): the driver must pick
    # the three-sweep pipeline and the i32 sweep by itself.  Uniform (+1 everywhere) and non-uniform gap costs.
    og = oracle.Graph.from_gfa_text(g.gfa())
    for tweak in ({"ACGTN": 1}, {"A": 2, "C": -3, "G": 1, "T": -6, "N": -2}):
        sm = api.create_score_matrix_i32(3, -5)
        for bases, v in tweak.items():
            for b in bases:
                sm[(b, "-")] = v
                sm[("-", b)] = v
        table = api._table_from_dict(sm)
        for mode, om in ((api.MODE_RECOMBINATION, oracle.M8_ABS), (api.MODE_PATHWISE, oracle.M4_ABS), (api.MODE_RECOMBINATION_SEMI, oracle.M9_ABS)):
            # (RG_SLOW_TESTS=1: the full sets of round 4 — 12 reads, 8 in the semiglobal mode, two literal comparisons per tweak;
            # the default run keeps the trimmed set, a nightly run restores the rest)
            reads = rd[:12 if SLOW else 8] if mode != api.MODE_RECOMBINATION_SEMI else [r[:170] for r in rd[:8 if SLOW else 6]]
            texts, _ = api.align_batch(gg, reads, None, mode=mode, score_matrix=sm)
            for i, r in enumerate(reads):
                assert texts[i] == og.align(om, r, name="read%d" % i, scores=table)[0], (tweak, mode, i)
        # and against the LITERAL restatement (delta-encoded matrices, pruned scan) on one read (the suite's time: ~10 s each)
        nlit = 2 if SLOW else 1
        texts, _ = api.align_batch(gg, rd[:nlit], None, mode=api.MODE_RECOMBINATION, score_matrix=sm)
        for i in range(nlit):
            assert texts[i] == og.align(oracle.M8_PRUNED, rd[i], name="read%d" % i, scores=table)[0]


def test_more_than_64_paths(oracle):
    """Path sets wider than one 64-bit word (P up to 256): groups whose members span several 64-path pages run as one
    alpha entry + continuation entries; packed and i32 sweeps, global and semiglobal, two- and three-sweep pipelines."""
    from recgraph_amd import api, synth
    for P, rows, plen, seed in ((70, 900, 120, 51), (130, 1400, 150, 52), (256, 1500, 90, 53)):
        g = synth.haplotype_graph(rows, P, path_len=plen, seed=seed)
        rd = synth.haplotype_reads(g, 20, length=plen, seed=seed + 100, mosaic_frac=0.5)
        rd += [g.path_sequence(P - 1)[:plen], g.path_sequence(64 if P > 64 else 0)[:plen // 2] + g.path_sequence(P - 2)[plen // 2:plen]]
        base = {}
        for mode, om in ((api.MODE_PATHWISE, oracle.M4_ABS), (api.MODE_RECOMBINATION, oracle.M8_ABS),
                         (api.MODE_PATHWISE_SEMI, oracle.M5_ABS), (api.MODE_RECOMBINATION_SEMI, oracle.M9_ABS)):
            reads = rd if mode in (api.MODE_PATHWISE, api.MODE_RECOMBINATION) else [r[:plen * 2 // 3] for r in rd[:10]]
            base[mode] = (reads, _check(oracle, g.gfa(), reads, mode, om))
        gg = api.Graph.from_gfa_text(g.gfa())
        for var in ("RG_SWEEP_I32", "RG_THREE_SWEEPS"):
            _switch(var, 1)
            for mode in (api.MODE_RECOMBINATION, api.MODE_RECOMBINATION_SEMI):
                reads, exp = base[mode]
                texts, _ = api.align_batch(gg, reads, ["r%d" % i for i in range(len(reads))], mode=mode)
                assert texts == exp, (P, var, mode)
            _switch(var, 0)
        _check(oracle, g.gfa(), rd[:3], api.MODE_RECOMBINATION, oracle.M8_PRUNED)      # the literal restatement
        tiny = ["GG", "A", "ACG", rd[0][:9]] + rd[:4]
        exp = _check(oracle, g.gfa(), tiny, api.MODE_RECOMBINATION, oracle.M8_ABS, R=0, r=0.1, B=0.8)
        for var in ("RG_SWEEP_I32", "RG_NO_FREC", "RG_THREE_SWEEPS"):
            _switch(var, 1)
            texts, _ = api.align_batch(gg, tiny, ["r%d" % i for i in range(len(tiny))], mode=api.MODE_RECOMBINATION, R=0, r=0.1, B=0.8)
            _switch(var, 0)
            assert texts == exp, (P, var)


def test_reads_longer_than_2047_bases(oracle):
    """Striped long reads: column stripes of 1024 (up to 8191 bases) or 2048, one wave per stripe in one workgroup,
    carries through LDS FIFOs (k_sweep / k_layer <C, true, true>), 3 to 7 stripes, mixed with short reads in the same batch."""
    from recgraph_amd import api, synth
    for plen, rows, P, nreads, seed in ((2600, 4200, 4, 5, 61), (5000, 7000, 3, 3, 62), (7000, 9000, 2, 2, 63)):
        g = synth.haplotype_graph(rows, P, path_len=plen, seed=seed)
        rd = synth.haplotype_reads(g, nreads, length=plen, seed=seed + 100, mosaic_frac=0.7)
        rd += [g.path_sequence(P - 1)[:plen - 37], g.path_sequence(0)[:300], "ACGT" * 3]
        _check(oracle, g.gfa(), rd, api.MODE_PATHWISE, oracle.M4_ABS)
        _check(oracle, g.gfa(), rd, api.MODE_RECOMBINATION, oracle.M8_ABS)
        if plen == 7000 and not SLOW:
            continue        # (the parameter variation and the semiglobal modes run at the two shorter lengths: the oracle's time; RG_SLOW_TESTS=1 runs them here too)
        _check(oracle, g.gfa(), rd[:3], api.MODE_RECOMBINATION, oracle.M8_ABS, R=1, r=0.5, B=0.7)
        semi = [r[:len(r) * 2 // 3] for r in rd[:3]]
        _check(oracle, g.gfa(), semi, api.MODE_PATHWISE_SEMI, oracle.M5_ABS)
        _check(oracle, g.gfa(), semi, api.MODE_RECOMBINATION_SEMI, oracle.M9_ABS)
        if plen == 2600:
            # the same batch at the other stripe widths (512 and 2048 columns per wave: 6 and 2 stripes)
            gg = api.Graph.from_gfa_text(g.gfa())
            names = ["r%d" % i for i in range(len(rd))]
            base = {m: api.align_batch(gg, rd, names, mode=m)[0] for m in (api.MODE_PATHWISE, api.MODE_RECOMBINATION)}
            for c in (8, 32):
                api.set_option("stripe_c", c)
                try:
                    for m in base:
                        assert api.align_batch(gg, rd, names, mode=m)[0] == base[m], (c, m)
                finally:
                    api.set_option("stripe_c", 0)
    # beyond 8 stripes of 1024 columns: 2048-column stripes
    g = synth.haplotype_graph(10500, 2, path_len=8400, seed=64)
    rd = synth.haplotype_reads(g, 1, length=8400, seed=164, mosaic_frac=1.0)
    _check(oracle, g.gfa(), rd, api.MODE_PATHWISE, oracle.M4_ABS)
    _check(oracle, g.gfa(), rd, api.MODE_RECOMBINATION, oracle.M8_ABS)


def test_striped_long_reads_retire_paths(oracle):
    """Path retirement ACROSS the stripes of a long read (round 5): every stripe publishes, per needed path, the maximum over
    its own columns; the decision is applied two evaluation points later by all stripes at the same record (their FIFOs
    carry one entry per row update: a stripe that skipped a record alone would hang the read).  Evaluation periods from 8 to
    256 records — short periods make the leading stripe wait for the slowest one at every point — stripes of 512 and 1024
    columns, against the oracle; and the retirement must actually bite."""
    from recgraph_amd import api, synth
    g = synth.haplotype_graph(5200, 8, path_len=2600, seed=71)
    rd = synth.haplotype_reads(g, 6, length=2600, seed=171, mosaic_frac=0.5)
    rd += [g.path_sequence(3)[:2300], g.path_sequence(0)[:700]]
    gg = api.Graph.from_gfa_text(g.gfa())
    names = ["r%d" % i for i in range(len(rd))]
    try:
        for shift, stripe_c in ((8, 0), (5, 0), (3, 0), (4, 8), (6, 8)):
            api.set_option("retire_shift", shift)
            api.set_option("stripe_c", stripe_c)
            _check(oracle, g.gfa(), rd, api.MODE_RECOMBINATION, oracle.M8_ABS)
            _check(oracle, g.gfa(), rd[:3], api.MODE_RECOMBINATION, oracle.M8_ABS, R=2, r=0.3, B=0.8)
        api.set_option("stripe_c", 0)
        api.set_option("retire_shift", 6)
        base = api.align_batch(gg, rd, names, mode=api.MODE_RECOMBINATION)[0]
        got = {}
        for key, val in (("on", 0), ("off", 1), ("forward_only", 2), ("reverse_only", 3)):
            api.set_option("no_retire", val)
            b = api.Batch(gg, rd, api.make_params(api.MODE_RECOMBINATION))
            b.run()
            b.fetch()
            got[key] = (b.cell_updates, b.cell_updates_performed)
            assert api.align_batch(gg, rd, names, mode=api.MODE_RECOMBINATION)[0] == base, key
        assert got["on"][0] == got["off"][0]
        assert got["off"][1] == got["off"][0]
        assert got["on"][1] < 0.8 * got["off"][1], got
        assert got["on"][1] < got["forward_only"][1] < got["off"][1], got
        assert got["on"][1] < got["reverse_only"][1] < got["off"][1], got
    finally:
        api.set_option("no_retire", 0)
        api.set_option("stripe_c", 0)
        api.set_option("retire_shift", 8)


def test_longest_supported_reads_eight_stripes_of_2048_columns(oracle):
    """A read near the 16 383-base limit: EIGHT stripes of 2048 columns in one workgroup (the launch with the largest LDS and
    register footprint of the library).  `-m 4` against the oracle; `-m 8` (81 s per read in the oracle) against `-m 4`: with
    a recombination cost no pair can pay for, its best score is the best single path's."""
    import re
    from recgraph_amd import api, synth
    g = synth.haplotype_graph(19000, 2, path_len=16000, seed=65)
    rd = synth.haplotype_reads(g, 1, length=16000, seed=165, mosaic_frac=1.0)
    assert 14336 < len(rd[0]) <= 16383
    t4 = _check(oracle, g.gfa(), rd, api.MODE_PATHWISE, oracle.M4_ABS)
    gg = api.Graph.from_gfa_text(g.gfa())
    t8, st = api.align_batch(gg, rd, ["r0"], mode=api.MODE_RECOMBINATION, R=100000)
    assert not any(st) and "recombination path" not in t8[0]
    assert int(re.search(r"score: (-?\d+)\t", t8[0]).group(1)) == int(re.search(r"score: (-?\d+)\t", t4[0]).group(1))
    t8d, st = api.align_batch(gg, rd, ["r0"], mode=api.MODE_RECOMBINATION)
    assert not any(st)


def test_wide_graphs_retire_paths_and_speculate(oracle):
    """Round 6: graphs with more than 64 paths run on the speculative bound, retire paths (several 64-bit words of `needed`,
    lead tables of several words, groups that span pages) and store direction words on demand like the narrow ones.  70, 130
    and 200 paths, block-built and random-walk graphs, against the oracle — with the retirement evaluated every 16 / 4 records
    (graphs of this size retire nothing at the default period), forward / reverse only, speculation off / zero margin / forced
    second pass, one-path picks, and without the always-stored edge rows; and the counters say that paths WERE retired."""
    from recgraph_amd import api, synth
    cases = [(synth.haplotype_graph(900, 70, path_len=150, seed=71), 150), (synth.haplotype_graph(1300, 130, path_len=180, seed=72), 180),
             (synth.haplotype_graph(1100, 200, path_len=140, seed=73), 140)]
    g4 = synth.random_dag_graph(170, 90, seed=74, max_jump=4, max_seg=9, similar=0.6)
    cases.append((g4, min(len(g4.path_sequence(k)) for k in range(90))))
    switches = ((), (("retire_shift", 4),), (("retire_shift", 2),), (("retire_shift", 4), ("no_retire", 2)), (("retire_shift", 4), ("no_retire", 3)),
                (("no_retire", 1),), (("no_spec", 1),), (("retire_shift", 4), ("spec_margin", 0)), (("spec_margin", -1000000),),
                (("retire_shift", 4), ("no_pick2", 1)), (("retire_shift", 3), ("dsel_edge", 1000)), (("no_dsel", 1),), (("retire_shift", 4), ("chunk_reads", 5)))
    defaults = {"spec_margin": SPEC_MARGIN_DEFAULT, "retire_shift": 8, "dsel_edge": 8}
    retired = 0
    for g, plen in cases:
        P = len(g.paths)
        rd = synth.haplotype_reads(g, 14, length=plen, seed=700 + P, mosaic_frac=0.6) + [g.path_sequence(P - 1)[:plen], g.path_sequence(65)[:plen]]
        gg = api.Graph.from_gfa_text(g.gfa())
        names = ["r%d" % i for i in range(len(rd))]
        base = _check(oracle, g.gfa(), rd, api.MODE_RECOMBINATION, oracle.M8_ABS)
        for combo in switches:
            try:
                for name, val in combo:
                    api.set_option(name, val)
                b = api.Batch(gg, rd, api.make_params(api.MODE_RECOMBINATION))
                b.run()
                b.fetch()
                texts = [b.gaf_text(i, names[i], i + 1) for i in range(len(rd))]
                if combo == (("retire_shift", 4),) and b.cell_updates_performed < 0.9 * b.cell_updates:
                    retired += 1
            finally:
                for name, _ in combo:
                    api.set_option(name, defaults.get(name, 0))
            assert texts == base, (P, combo)
    assert retired >= 2, retired
    # reads of more than 1 023 bases on a wide graph (32 columns per lane: no register runs there, gather runs whose member
    # passes iterate the pages of the group — the -m 4 variant included: a fuzz campaign found it reading an empty member set)
    g = synth.haplotype_graph(3300, 70, path_len=1150, seed=75)
    rd = synth.haplotype_reads(g, 6, length=1150, seed=775, mosaic_frac=0.5) + [g.path_sequence(69)[:1150]]
    _check(oracle, g.gfa(), rd, api.MODE_PATHWISE, oracle.M4_ABS)
    _check(oracle, g.gfa(), rd, api.MODE_RECOMBINATION, oracle.M8_ABS)
