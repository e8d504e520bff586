"""CPU: pins the oracle against every known-answer vector the reference's own unit tests hold
(tests/golden/reference_unit_vectors.json), the hand-derived m4/m8 vectors, and checks that the two
structurally different restatements of the pathwise modes agree."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
VEC = json.load(open(os.path.join(HERE, "golden", "reference_unit_vectors.json")))


def _scores(oracle, d):
    return oracle.scores_from_dict({(k[0], k[1]): v for k, v in d.items()})


def _preds(d):
    return {int(k): v for k, v in d.items()}


@pytest.mark.parametrize("v", VEC["global_abpoa_scalar"], ids=lambda v: v["ref"])
def test_m0_scalar_reference_scores(oracle, v):
    g = oracle.Graph.lnz_literal(v["lnz"], _preds(v["preds"]))
    out, score, panic, _ = g.align(oracle.M0_SCALAR, v["read"], idx=0, scores=_scores(oracle, v["scores"]), bta=v["bta"])
    assert not panic and score == v["score"]


@pytest.mark.parametrize("v", VEC["gap_global_abpoa"], ids=lambda v: v["ref"])
def test_m2_reference_scores(oracle, v):
    g = oracle.Graph.lnz_literal(v["lnz"], _preds(v["preds"]))
    out, score, panic, _ = g.align(oracle.M2, v["read"], idx=0, scores=_scores(oracle, v["scores"]), o=v["o"], e=v["e"],
                                   bta=v["bta"])
    assert not panic and score == v["score"]


@pytest.mark.parametrize("v", VEC["local_poa"], ids=lambda v: v["ref"])
def test_m1_reference_scores(oracle, v):
    """local_poa.rs:300-378 (the tests call the scalar exec; the AVX2 restatement must agree on them)."""
    g = oracle.Graph.lnz_literal(v["lnz"], _preds(v["preds"]))
    for mode in (oracle.M1_SCALAR, oracle.M1_SIMD):
        out, score, panic, _ = g.align(mode, v["read"], idx=0, scores=_scores(oracle, v["scores"]))
        assert not panic and score == v["score"]


@pytest.mark.parametrize("v", VEC["gap_local_poa"], ids=lambda v: v["ref"])
def test_m3_reference_scores(oracle, v):
    """gap_local_poa.rs:186-270."""
    g = oracle.Graph.lnz_literal(v["lnz"], _preds(v["preds"]))
    out, score, panic, _ = g.align(oracle.M3, v["read"], idx=0, scores=_scores(oracle, v["scores"]), o=v["o"], e=v["e"])
    assert not panic and score == v["score"]


def _smith_waterman(ref, read, m, x, gap):
    H = [[0] * (len(read) + 1) for _ in range(len(ref) + 1)]
    best = 0
    for i in range(1, len(ref) + 1):
        for j in range(1, len(read) + 1):
            s = m if ref[i - 1] == read[j - 1] else x
            H[i][j] = max(0, H[i - 1][j - 1] + s, H[i - 1][j] + gap, H[i][j - 1] + gap)
            best = max(best, H[i][j])
    return best


def _gotoh_local(ref, read, m, x, o, e):
    NEG = -10 ** 9
    n, w = len(ref), len(read)
    H = [[0] * (w + 1) for _ in range(n + 1)]
    X = [[0] * (w + 1) for _ in range(n + 1)]   # the reference keeps 0 on the borders of x / y
    Y = [[0] * (w + 1) for _ in range(n + 1)]
    best = 0
    for i in range(1, n + 1):
        for j in range(1, w + 1):
            X[i][j] = max(X[i][j - 1] + e, H[i][j - 1] + o + e)
            Y[i][j] = max(Y[i - 1][j] + e, H[i - 1][j] + o + e)
            s = m if ref[i - 1] == read[j - 1] else x
            H[i][j] = max(0, H[i - 1][j - 1] + s, X[i][j], Y[i][j])
            best = max(best, H[i][j])
    return best


def test_local_modes_equal_smith_waterman_on_a_chain(oracle):
    """Independent check: on a single-segment graph local POA is Smith-Waterman (m1) / Gotoh with zero borders (m3)."""
    rng = np.random.default_rng(5)
    for t in range(40):
        ref = "".join("ACGT"[int(k)] for k in rng.integers(0, 4, size=int(rng.integers(2, 40))))
        read = "".join("ACGT"[int(k)] for k in rng.integers(0, 4, size=int(rng.integers(1, 30))))
        if t % 2:
            a = int(rng.integers(0, len(ref) - 1))
            read = read[:5] + ref[a:a + 12] + read[5:]
        g = oracle.Graph.from_gfa_text("S\t1\t%s\n" % ref, want_path=False)
        sc = oracle.scores_match_mis(2, -4)            # i32 variant: gaps cost -8
        scf = oracle.scores_match_mis(2, -4, True)     # f32 variant: gaps cost -4
        assert g.align(oracle.M1_SCALAR, read, idx=0, scores=sc)[1] == _smith_waterman(ref, read, 2, -4, -8)
        # the AVX2 flavour never clamps the scalar tail of a multi-predecessor row (local_poa.rs:128-156; row 1 always is
        # one), so it is Smith-Waterman only when the read leaves no tail: (n + 1) % 8 == 1
        r8 = (read * 8)[:max(8, len(read) // 8 * 8)]
        assert g.align(oracle.M1_SIMD, r8, idx=0, scores=scf)[1] == _smith_waterman(ref, r8, 2, -4, -4)
        assert g.align(oracle.M3, read, idx=0, scores=sc, o=-4, e=-2)[1] == _gotoh_local(ref, read, 2, -4, -4, -2)


def test_m2_with_o0_equals_m0_scalar(oracle):
    """The reference's own cross-check idea (gap_global_abpoa.rs:642): o = 0 and e = gap score."""
    rng = np.random.default_rng(1)
    g = oracle.Graph.lnz_literal("$AACAAAF", {1: [0], 3: [2], 4: [2], 5: [3, 4], 7: [6]})
    sc = oracle.scores_from_dict({("A", "A"): 1, ("C", "C"): 1, ("A", "C"): -1, ("C", "A"): -1, ("A", "-"): -1,
                                  ("-", "A"): -1, ("C", "-"): -1, ("-", "C"): -1})
    for _ in range(50):
        rd = "".join("AC"[int(x)] for x in rng.integers(0, 2, size=int(rng.integers(1, 9))))
        a = g.align(oracle.M0_SCALAR, rd, idx=0, scores=sc, bta=20)
        b = g.align(oracle.M2, rd, idx=0, scores=sc, o=0, e=-1, bta=20)
        assert a[1] == b[1], rd


@pytest.mark.parametrize("v", VEC["graph_struct"], ids=lambda v: v["ref"])
def test_lnz_graph_construction(oracle, v):
    g = oracle.Graph.from_gfa_text(v["gfa"], want_path=False)
    if "lnz" in v:
        assert g.dump(0) == v["lnz"]
        nwp = g.dump(1)
        for i in v["nwp_set"]:
            assert nwp[i] == "1"
        preds = dict(x.split(":") for x in g.dump(2).strip(";").split(";"))
        for k, p in v["preds"].items():
            assert [int(t) for t in preds[k].split(",")] == p
    if "handle_index_of_row" in v:
        hofp = g.dump(3).split(",")
        ids = sorted({int(x) for x in hofp[1:]})
        for row, idx in v["handle_index_of_row"].items():
            assert ids.index(int(hofp[int(row)])) == idx


@pytest.mark.parametrize("v", VEC["path_graph"], ids=lambda v: v["ref"])
def test_path_graph_construction(oracle, v):
    g = oracle.Graph.from_gfa_text(v["gfa"])
    rows = g.dump(13).strip(";").split(";")
    assert len(rows[0]) == v["paths_number"]
    if "lnz" in v:
        assert g.dump(10) == v["lnz"]
    for i in v.get("nwp_set", []):
        assert g.dump(11)[i] == "1"
    for row, bits in v["paths_nodes"].items():
        assert rows[int(row)] == bits
    if "pred_hash" in v:
        ph = {}
        for ent in g.dump(12).strip(";").split(";"):
            r, es = ent.split(":")
            ph[r] = dict(e.split("=") for e in es.split(","))
        for r, es in v["pred_hash"].items():
            assert ph[r] == es


@pytest.mark.parametrize("v", VEC["score_matrix"], ids=lambda v: v["ref"])
def test_score_matrices(oracle, v):
    if "mtx" in v:
        t = oracle.scores_from_mtx(open(os.path.join(HERE, "golden", v["mtx"] + ".mtx")).read())
    else:
        t = oracle.scores_match_mis(v["m"], v["x"])
    al = oracle.ALPHABET
    for k, val in v["entries"].items():
        assert t[al.index(k[0]) * 6 + al.index(k[1])] == val
    for k in v["absent"]:
        assert t[al.index(k[0]) * 6 + al.index(k[1])] == oracle.lib().orc_missing_value()


@pytest.mark.parametrize("v", VEC["hand_derived"], ids=lambda v: "m%d %s" % (v["mode"], v["read"]))
def test_hand_derived_pathwise_vectors(oracle, v):
    g = oracle.Graph.from_gfa_text(v["gfa"])
    modes = (oracle.M4, oracle.M4_ABS) if v["mode"] == 4 else (oracle.M8, oracle.M8_PRUNED, oracle.M8_ABS)
    for m in modes:
        assert g.align(m, v["read"], name=v["name"])[0] == v["gaf"]
    if "dfs" in v:
        assert [int(x) for x in g.dump(18).split(",")] == v["dfs"]
        assert [int(x) for x in g.dump(19).split(",")] == v["dfe"]
        assert [int(x) for x in g.dump(14).split(",")] == v["alphas"]


def test_f32_path_cell_decoding_is_exact_below_2_20(oracle):
    """gaf_output.rs:783-786 decodes `pred + 0.1|0.2|0.3` through Display + split('.'): exact for every
    row below 2^20, which is the bound rg_batch_create enforces for -m 0."""
    assert oracle.lib().orc_f32_cell_roundtrip_limit(1 << 20) == -1
    assert oracle.lib().orc_f32_cell_roundtrip_limit((1 << 20) + 8) == 1 << 20


def test_literal_and_absolute_restatements_agree(oracle):
    """Pin (ii) of the oracle header: delta-encoded transliteration (unpruned O(L^2 n) search) vs the
    absolute-score formulation, on randomised small graphs and reads."""
    from recgraph_amd import synth
    rng = np.random.default_rng(5)
    for seed in range(6):
        sg = synth.haplotype_graph(int(rng.integers(60, 140)), int(rng.integers(2, 7)), path_len=int(rng.integers(14, 30)),
                                   seed=100 + seed)
        g = oracle.Graph.from_gfa_text(sg.gfa())
        n = len(sg.path_sequence(0))
        reads = synth.haplotype_reads(sg, 6, length=max(4, n - 2), seed=seed, mosaic_frac=0.5)
        reads += ["ACGT", sg.path_sequence(1)[:n // 2] + sg.path_sequence(0)[n // 2:]]
        for rd in reads:
            a = g.align(oracle.M4, rd)[0]
            assert a == g.align(oracle.M4_ABS, rd)[0]
            for kw in ({}, {"R": 0, "r": 0.0}, {"R": 2, "r": 0.7, "B": 0.5}):
                b = g.align(oracle.M8, rd, **kw)[0]
                assert b == g.align(oracle.M8_PRUNED, rd, **kw)[0]
                assert b == g.align(oracle.M8_ABS, rd, **kw)[0]


def test_restatements_agree_on_random_walk_graphs(oracle):
    """The same pin on graphs that are not allele blocks (`synth.random_dag_graph`: nested / overlapping bubbles, paths
    that part ways after a shared segment): the GPU parity tests on such graphs compare with the absolute forms."""
    from recgraph_amd import synth
    for seed, nseg, P, kw in ((1, 12, 3, {}), (2, 16, 5, {"max_jump": 3}), (3, 10, 4, {"max_seg": 4, "similar": 0.8}), (4, 20, 2, {"max_jump": 6})):
        sg = synth.random_dag_graph(nseg, P, seed=400 + seed, **kw)
        g = oracle.Graph.from_gfa_text(sg.gfa())
        n = min(len(sg.path_sequence(k)) for k in range(P))
        reads = synth.haplotype_reads(sg, 4, length=max(4, min(n, 40)), seed=seed, mosaic_frac=0.6) + ["ACGT", sg.path_sequence(P - 1)[:40]]
        for rd in reads:
            assert g.align(oracle.M4, rd)[0] == g.align(oracle.M4_ABS, rd)[0]
            assert g.align(oracle.M5, rd[:len(rd) * 2 // 3 + 1])[0] == g.align(oracle.M5_ABS, rd[:len(rd) * 2 // 3 + 1])[0]
            for kw2 in ({}, {"R": 2, "r": 0.7, "B": 0.5}):
                b = g.align(oracle.M8, rd, **kw2)[0]
                assert b == g.align(oracle.M8_PRUNED, rd, **kw2)[0]
                assert b == g.align(oracle.M8_ABS, rd, **kw2)[0]
                assert g.align(oracle.M9_PRUNED, rd, **kw2)[0] == g.align(oracle.M9_ABS, rd, **kw2)[0]


def test_semiglobal_restatements_agree(oracle):
    """-m 5 / -m 9: literal vs absolute-form restatement, unpruned vs pruned search."""
    from recgraph_amd import synth
    rng = np.random.default_rng(6)
    for seed in range(5):
        sg = synth.haplotype_graph(int(rng.integers(60, 140)), int(rng.integers(2, 7)), path_len=int(rng.integers(14, 30)),
                                   seed=300 + seed)
        g = oracle.Graph.from_gfa_text(sg.gfa())
        n = len(sg.path_sequence(0))
        reads = synth.haplotype_reads(sg, 4, length=max(4, n - 2), seed=seed, mosaic_frac=0.5)
        reads += ["ACGT", sg.path_sequence(1)[n // 3:2 * n // 3], sg.path_sequence(0)[:n // 2]]
        for rd in reads:
            assert g.align(oracle.M5, rd)[0] == g.align(oracle.M5_ABS, rd)[0]
            for kw in ({}, {"R": 0, "r": 0.0}, {"R": 2, "r": 0.7, "B": 0.5}):
                b = g.align(oracle.M9, rd, **kw)[0]
                assert b == g.align(oracle.M9_PRUNED, rd, **kw)[0]
                assert b == g.align(oracle.M9_ABS, rd, **kw)[0]


def test_example_data_restatements_agree(oracle, example_gfa, example_reads):
    names, reads = example_reads
    g = oracle.Graph.from_gfa_text(example_gfa)
    for i in (0, 7, 19):
        assert g.align(oracle.M4, reads[i], name=names[i])[0] == g.align(oracle.M4_ABS, reads[i], name=names[i])[0]
        assert g.align(oracle.M8_PRUNED, reads[i], name=names[i])[0] == g.align(oracle.M8_ABS, reads[i], name=names[i])[0]


def test_oracle_f32_display_against_numpy_shortest_form(oracle):
    """The checker's own `{}`-of-f32 formatter (orc_common.cpp, deliberately a different algorithm from the product's) against
    numpy's Dragon4 shortest-unique positional text: every k/10 of the score range, the penalty arithmetic, random patterns."""
    import ctypes as C
    import numpy as np
    tot = np.arange(-3000, 3001, dtype=np.float32)
    vals = [(tot - (np.float32(4) + np.float32(0.1) * np.float32(d))).astype(np.float32) for d in (0, 1, 3, 17, 999)]
    vals.append((np.arange(-40000, 40001, 7, dtype=np.float32) / np.float32(10)).astype(np.float32))
    rb = np.random.default_rng(9).integers(0, 1 << 32, size=20000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    vals.append(rb[np.isfinite(rb)])
    vals.append(np.array([0.0, -0.0, 1e-45, 3.4028235e38, 1.1754944e-38, 16777216.0, 0.1, 1e7, 1e-7, 2.3621053e30], dtype=np.float32))
    v = np.concatenate(vals).astype(np.float32)
    buf = C.create_string_buffer(128)
    f = oracle.lib().orc_f32_display
    for x, b in zip(v, v.view(np.uint32)):
        assert f(int(b), buf, 128) > 0
        assert buf.value.decode() == np.format_float_positional(x, unique=True, trim="-"), (x, buf.value)
