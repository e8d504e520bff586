"""Local POA (-m 1 in both flavours, -m 3) parity on the GPU, through the C ABI, vs the oracle."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
VEC = json.load(open(os.path.join(HERE, "golden", "reference_unit_vectors.json")))


def _sd(d):
    return {(k[0], k[1]): v for k, v in d.items()}


@pytest.mark.parametrize("v", VEC["local_poa"], ids=lambda v: v["ref"])
def test_reference_local_vectors(v):
    """local_poa.rs:300-378, written the way the reference writes them."""
    from recgraph_amd import api
    g = api.Graph.from_lnz(v["lnz"], {int(k): p for k, p in v["preds"].items()})
    for scalar in (True, False):
        score, gaf = api.local_poa_exec(["$"] + list(v["read"]), ("seq", 0), g, _sd(v["scores"]), scalar=scalar)
        assert score == v["score"] and gaf is None


@pytest.mark.parametrize("v", VEC["gap_local_poa"], ids=lambda v: v["ref"])
def test_reference_gap_local_vectors(v):
    """gap_local_poa.rs:186-270."""
    from recgraph_amd import api
    g = api.Graph.from_lnz(v["lnz"], {int(k): p for k, p in v["preds"].items()})
    score, gaf = api.gap_local_poa_exec(["$"] + list(v["read"]), ("test", 0), g, _sd(v["scores"]), v["o"], v["e"])
    assert score == v["score"] and gaf is None


MODES = None


def _modes(oracle):
    from recgraph_amd import api
    return ((api.MODE_LOCAL_POA, oracle.M1_SIMD), (api.MODE_LOCAL_POA_SCALAR, oracle.M1_SCALAR),
            (api.MODE_GAP_LOCAL_POA, oracle.M3))


def _compare(oracle, gfa, reads, mode, omode, **kw):
    from recgraph_amd import api
    og = oracle.Graph.from_gfa_text(gfa, want_path=False)
    g = api.Graph.from_gfa_text(gfa)
    names = ["r%d" % i for i in range(len(reads))]
    p = api.make_params(mode, **kw)
    b = api.Batch(g, reads, p)
    b.run()
    b.fetch()
    okw = {k: v for k, v in kw.items() if k != "score_matrix"}
    if "score_matrix" in kw:
        okw["scores"] = oracle.scores_from_dict(kw["score_matrix"])
    bad = []
    nonzero = 0
    for i, rd in enumerate(reads):
        exp, score, panic, _ = og.align(omode, rd, name=names[i], idx=i + 1, **okw)
        assert not panic
        nonzero += score > 0
        got = b.gaf_text(i, names[i], i + 1)
        if got != exp or b.score(i) != score:
            bad.append((i, rd[:30], b.score(i), score, got[:300], exp[:300]))
    assert not bad, (len(bad), bad[:2])
    return nonzero


def _mutate(s, rng, rate=25):
    s = list(s)
    for _ in range(max(1, len(s) // rate)):
        s[int(rng.integers(0, len(s)))] = "ACGT"[int(rng.integers(0, 4))]
    if len(s) > 12 and rng.integers(0, 2):
        del s[5:5 + int(rng.integers(1, 4))]
    if len(s) > 8 and rng.integers(0, 2):
        s[3:3] = list("GTG"[:int(rng.integers(1, 4))])
    return "".join(s)


def _reads_for(sg, rng, lengths):
    """Substrings of path walks (start anywhere: local alignment), mutated, flanked by unrelated bases."""
    reads = []
    for k, n in enumerate(lengths):
        walk = sg.path_sequence(k % len(sg.paths))
        a = int(rng.integers(0, max(1, len(walk) - n)))
        core = _mutate(walk[a:a + n], rng)
        if k % 3 == 0:
            junk = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=int(rng.integers(1, 12))))
            core = (junk + core)[:n] if k % 2 else (core + junk)[-n:]
        reads.append(core[:n] if len(core) >= n else core + "A" * (n - len(core)))
    reads.append("".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=40)))   # unrelated read
    reads += ["A", "C", "NN", "ACG", "N" * 9]
    return reads


def test_local_synthetic_chain(oracle):
    from recgraph_amd import synth
    rng = np.random.default_rng(31)
    sg = synth.linear_graph(400, seed=3)
    # W = n + 1: cover the AVX2 flavour's chunk/tail split (W % 8 == 0, 1, 2 ...) and the 64-column chunk carries
    lengths = [7, 8, 9, 15, 16, 17, 23, 31, 32, 33, 62, 63, 64, 65, 66, 100, 127, 128, 129, 150, 191, 192, 193, 250]
    reads = _reads_for(sg, rng, lengths)
    for mode, omode in _modes(oracle):
        assert _compare(oracle, sg.gfa(), reads, mode, omode) > len(lengths) // 2


def test_local_haplotype_graph(oracle):
    """Bubbles: multi-predecessor rows exercise the `first = false` quirk and the unclamped AVX2 tail."""
    from recgraph_amd import api, synth
    rng = np.random.default_rng(32)
    sg = synth.haplotype_graph(400, 6, path_len=150, seed=5)
    lengths = [int(x) for x in rng.integers(5, 180, size=48)] + [8, 16, 24, 64, 72]
    reads = _reads_for(sg, rng, lengths)
    for mode, omode in _modes(oracle):
        _compare(oracle, sg.gfa(), reads, mode, omode)
    # score variants: unit scores (ties everywhere), a substitution matrix, other gap penalties
    unit = {(a, b): (1 if a == b else -1) for a in "ACGTN-" for b in "ACGTN-" if (a, b) != ("-", "-")}
    for mode, omode in _modes(oracle):
        _compare(oracle, sg.gfa(), reads[:40], mode, omode, score_matrix=unit)
    hox = api.create_score_matrix_i32(matrix_file_path=os.path.join(HERE, "golden", "HOXD70.mtx"))
    for mode, omode in _modes(oracle):
        _compare(oracle, sg.gfa(), reads[:40], mode, omode, score_matrix=hox)
    for o, e in ((0, -3), (-10, -6), (-1, -1), (-4, 0)):
        _compare(oracle, sg.gfa(), reads[:40], api.MODE_GAP_LOCAL_POA, oracle.M3, o=o, e=e)
        _compare(oracle, sg.gfa(), reads[:40], api.MODE_GAP_LOCAL_POA, oracle.M3, o=o, e=e, score_matrix=unit)


def test_local_single_base_nodes(oracle):
    """Every row is its own node (all rows multi-predecessor): the AVX2 flavour's tail cells are never clamped."""
    rng = np.random.default_rng(33)
    segs = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=30))
    gfa = "".join("S\t%d\t%s\n" % (i + 1, c) for i, c in enumerate(segs))
    gfa += "".join("L\t%d\t+\t%d\t+\t0M\n" % (i + 1, i + 2) for i in range(len(segs) - 1))
    gfa += "".join("L\t%d\t+\t%d\t+\t0M\n" % (i + 1, i + 3) for i in range(0, len(segs) - 2, 3))
    reads = [segs[a:a + n] for a in (0, 3, 11) for n in (1, 2, 5, 7, 8, 12, 19)]
    reads += ["".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=int(n))) for n in rng.integers(1, 20, size=30)]
    for mode, omode in _modes(oracle):
        _compare(oracle, gfa, reads, mode, omode)


def test_local_example_data(oracle, example_gfa, example_reads):
    names, reads = example_reads
    rd = reads[:24] + [reads[0][40:110], reads[1][:33], reads[2][75:]]
    for mode, omode in _modes(oracle):
        assert _compare(oracle, example_gfa, rd, mode, omode) == len(rd)


def test_local_cli(oracle, tmp_path, example_gfa, example_reads, capsys):
    """main.rs:108-168 / :196-226 through the command line mirror."""
    from recgraph_amd import cli
    names, reads = example_reads
    gp, rp = tmp_path / "g.gfa", tmp_path / "r.fa"
    gp.write_text(example_gfa)
    rp.write_text("".join(">%s\n%s\n" % (names[i], reads[i]) for i in range(6)))
    og = oracle.Graph.from_gfa_text(example_gfa, want_path=False)
    for m, omode, extra in ((1, oracle.M1_SIMD, []), (1, oracle.M1_SCALAR, ["--scalar"]), (3, oracle.M3, [])):
        cli.main([str(rp), str(gp), "-m", str(m)] + extra)
        out = capsys.readouterr().out
        exp = "".join(og.align(omode, reads[i], name=names[i], idx=i + 1)[0] for i in range(6))
        assert out == exp
