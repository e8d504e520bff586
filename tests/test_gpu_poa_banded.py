"""-m 2 (affine gaps) and scalar -m 0 parity on the GPU, through the C ABI, vs the oracle."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
VEC = json.load(open(os.path.join(HERE, "golden", "reference_unit_vectors.json")))


def _sd(d):
    return {(k[0], k[1]): v for k, v in d.items()}


@pytest.mark.parametrize("v", VEC["global_abpoa_scalar"], ids=lambda v: v["ref"])
def test_reference_scalar_m0_vectors(v):
    """The reference's own unit tests, written the way they are written there (global_abpoa.rs:576-755)."""
    from recgraph_amd import api
    g = api.Graph.from_lnz(v["lnz"], {int(k): p for k, p in v["preds"].items()})
    score, gaf = api.global_abpoa_exec(["$"] + list(v["read"]), ("test", 0), g, _sd(v["scores"]), v["bta"])
    assert score == v["score"] and gaf is None


@pytest.mark.parametrize("v", VEC["gap_global_abpoa"], ids=lambda v: v["ref"])
def test_reference_gap_vectors(v):
    """gap_global_abpoa.rs:464-757."""
    from recgraph_amd import api
    g = api.Graph.from_lnz(v["lnz"], {int(k): p for k, p in v["preds"].items()})
    score, gaf = api.gap_global_abpoa_exec(["$"] + list(v["read"]), ("test", 0), g, _sd(v["scores"]), v["o"], v["e"], v["bta"])
    assert score == v["score"] and gaf is None


def _compare(oracle, gfa, reads, mode, omode, **kw):
    from recgraph_amd import api
    og = oracle.Graph.from_gfa_text(gfa, want_path=False)
    g = api.Graph.from_gfa_text(gfa)
    names = ["r%d" % i for i in range(len(reads))]
    texts, status = api.align_batch(g, reads, names, mode=mode, **kw)
    bad, panics, warns = [], 0, 0
    for i, rd in enumerate(reads):
        exp, score, panic, _ = og.align(omode, rd, name=names[i], idx=i + 1, **kw)
        if panic:
            panics += 1
            if not status[i] & api.READ_WOULD_PANIC:
                bad.append((i, "expected panic status", status[i]))
            continue
        warns += "Band length" in exp
        if texts[i] != exp:
            bad.append((i, rd[:30], texts[i][:400], exp[:400]))
    assert not bad, (len(bad), bad[:2])
    return panics, warns


def _reads_for(sg, n, length, rng):
    from recgraph_amd import synth
    walk = sg.path_sequence(0)
    reads = synth.substring_reads(sg, n, length, seed=int(rng.integers(1, 1000)))
    reads += [walk[:length], walk[:length // 2], walk[1:length + 1], walk[:3], walk[:1], walk, walk[len(walk) // 3:]]
    # reads anchored at the source with errors
    for k in range(12):
        s = list(sg.path_sequence(k % len(sg.paths))[:int(rng.integers(20, length + 1))])
        for _ in range(len(s) // 25):
            s[int(rng.integers(0, len(s)))] = "ACGT"[int(rng.integers(0, 4))]
        if k % 3 == 0 and len(s) > 10:
            del s[5:8]
        if k % 4 == 0:
            s[3:3] = list("GGG")
        reads.append("".join(s))
    return reads


def test_m2_synthetic(oracle):
    from recgraph_amd import api, synth
    rng = np.random.default_rng(21)
    sg = synth.linear_graph(500, seed=9)
    reads = _reads_for(sg, 64, 120, rng)
    for kw in ({}, {"o": -4, "e": -2, "b": 3.0}, {"o": 0, "e": -3}, {"o": -10, "e": -6, "bta": 12}, {"o": -2, "e": -1, "bta": 1},
               {"bta": 600}):
        _compare(oracle, sg.gfa(), reads, api.MODE_GAP_POA, oracle.M2, **kw)


def test_m0_scalar_synthetic(oracle):
    from recgraph_amd import api, synth
    rng = np.random.default_rng(22)
    sg = synth.linear_graph(500, seed=10)
    reads = _reads_for(sg, 64, 120, rng)
    for kw in ({}, {"b": 3.0}, {"bta": 12}, {"bta": 1}, {"bta": 600}):
        _compare(oracle, sg.gfa(), reads, api.MODE_GLOBAL_POA_SCALAR, oracle.M0_SCALAR, **kw)


def test_example_data(oracle, example_gfa, example_reads):
    from recgraph_amd import api
    names, reads = example_reads
    segs, paths = {}, []
    for line in example_gfa.splitlines():
        f = line.split("\t")
        if f[0] == "S":
            segs[f[1]] = f[2]
        elif f[0] == "P":
            paths.append("".join(segs[s[:-1]] for s in f[2].split(",")))
    rd = reads[:20] + [p[:k] for p in paths[:6] for k in (60, 200, len(p))]
    _compare(oracle, example_gfa, rd, api.MODE_GAP_POA, oracle.M2)
    _compare(oracle, example_gfa, rd, api.MODE_GAP_POA, oracle.M2, bta=40)
    _compare(oracle, example_gfa, rd, api.MODE_GLOBAL_POA_SCALAR, oracle.M0_SCALAR)
    _compare(oracle, example_gfa, rd, api.MODE_GLOBAL_POA_SCALAR, oracle.M0_SCALAR, bta=40)


def test_c3_sample(oracle):
    from recgraph_amd import api, synth
    sg, reads, _ = synth.make_config("C3", n_reads=96)
    walk = sg.path_sequence(0)
    reads += [walk[:500], walk[:250]]
    _compare(oracle, sg.gfa(), reads, api.MODE_GAP_POA, oracle.M2)
