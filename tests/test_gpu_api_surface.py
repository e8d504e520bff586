"""The reference's library surface (api.rs:11-164) executed end to end on the HIP path: the four ``align_*`` wrappers with
their DEFAULTS (f32 matrix whose gap entries equal the mismatch score, score_matrix.rs:52-66; ``o = -10, e = -6``;
``bases_to_add = (len as f32 * 0.1) as usize``; name ("no_name", 1)) and with explicit arguments, each returned
``GAFStruct`` against the oracle run on the same values."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOXD70 = os.path.join(ROOT, "tests", "golden", "HOXD70.mtx")


def _bta(read, frac):
    return int(np.float32(len(read)) * np.float32(frac))       # api.rs:22,58 — all in f32, then `as usize`


def _expect(api, text):
    """What ``alignment.1.unwrap()`` holds for the stdout text the oracle produced (last line = the record)."""
    if "band not enough" in text:
        return api.GAFStruct()
    return api.GAFStruct.from_line(text.rstrip("\n").split("\n")[-1])


def _table(oracle, d):
    return oracle.scores_from_dict({k: int(v) for k, v in d.items()})


@pytest.fixture(scope="module")
def setup(oracle, example_gfa, example_reads):
    from recgraph_amd import api
    names, reads = example_reads
    return api, api.Graph.from_gfa_text(example_gfa), oracle.Graph.from_gfa_text(example_gfa), reads[:10]


def test_align_global_no_gap_defaults_and_arguments(oracle, setup):
    api, g, og, reads = setup
    dflt = oracle.scores_match_mis(2, -4, f32_variant=True)
    assert dflt[0 * 6 + 5] == -4 and dflt[5 * 6 + 0] == -4           # gap = x, not 2x (score_matrix.rs:52-66)
    hits = 0
    for rd in reads:
        got = api.align_global_no_gap(rd, g)
        exp = og.align(oracle.M0_SIMD, rd, name="no_name", idx=1, scores=dflt, bta=_bta(rd, 0.1))[0]
        assert got == _expect(api, exp)
        if "band not enough" not in exp:
            hits += 1
            assert got.to_string() == exp.rstrip("\n").split("\n")[-1]
            assert got.query_name == "no_name"
    assert hits >= 5
    sm = api.create_score_matrix_f32(3, -5)                            # api.rs:153-164: i32 matrix as f32, gap = 2x
    assert sm[("A", "-")] == -10.0
    for rd in reads[:5]:
        got = api.align_global_no_gap(rd, g, sequence_name=("q7", 7), score_matrix=sm, bases_to_add=0.5)
        exp = og.align(oracle.M0_SIMD, rd, name="q7", idx=7, scores=_table(oracle, sm), bta=_bta(rd, 0.5))[0]
        assert got == _expect(api, exp)
        assert got.query_name == "q7"
    with pytest.raises(Exception):
        api.align_global_no_gap(reads[0], g, sequence_name=("x", 0))   # alignment.1 is None: unwrap panics (api.rs:38)


def test_align_global_gap_defaults_and_arguments(oracle, setup):
    api, g, og, reads = setup
    dflt = oracle.scores_match_mis(2, -4)
    for rd in reads:
        got = api.align_global_gap(rd, g)
        exp = og.align(oracle.M2, rd, name="no_name", idx=1, scores=dflt, o=-10, e=-6, bta=_bta(rd, 0.1))[0]
        assert got == _expect(api, exp)
        assert got.to_string() == exp.rstrip("\n").split("\n")[-1]
    hox = api.create_score_matrix_i32(matrix_file_path=HOXD70)         # score_matrix.rs:67-105: gap entries -200
    assert hox[("A", "-")] == -200 and hox[("G", "T")] != hox[("T", "G")]
    for rd in reads[:5]:
        got = api.align_global_gap(rd, g, sequence_name=("hox", 3), score_matrix=hox, bases_to_add=0.3, o=-400, e=-30)
        exp = og.align(oracle.M2, rd, name="hox", idx=3, scores=_table(oracle, hox), o=-400, e=-30, bta=_bta(rd, 0.3))[0]
        assert got == _expect(api, exp)
    for rd in reads[:3]:
        got = api.align_global_gap(rd, g, o=-7, e=-3)
        exp = og.align(oracle.M2, rd, name="no_name", idx=1, scores=dflt, o=-7, e=-3, bta=_bta(rd, 0.1))[0]
        assert got == _expect(api, exp)


def test_align_local_no_gap_defaults_and_arguments(oracle, setup):
    api, g, og, reads = setup
    dflt = oracle.scores_match_mis(2, -4, f32_variant=True)
    for rd in reads:
        got = api.align_local_no_gap(rd, g)
        exp = og.align(oracle.M1_SIMD, rd, name="no_name", idx=1, scores=dflt)[0]
        assert got == _expect(api, exp)
        assert got.to_string() == exp.rstrip("\n").split("\n")[-1]
    sm = api.create_score_matrix_f32(5, -3)
    for rd in reads[:5]:
        sub = rd[20:110]
        got = api.align_local_no_gap(sub, g, sequence_name=("loc", 12), score_matrix=sm)
        exp = og.align(oracle.M1_SIMD, sub, name="loc", idx=12, scores=_table(oracle, sm))[0]
        assert got == _expect(api, exp)


def test_align_local_gap_defaults_and_arguments(oracle, setup):
    api, g, og, reads = setup
    dflt = oracle.scores_match_mis(2, -4)
    for rd in reads:
        got = api.align_local_gap(rd, g)
        exp = og.align(oracle.M3, rd, name="no_name", idx=1, scores=dflt, o=-10, e=-6)[0]
        assert got == _expect(api, exp)
        assert got.to_string() == exp.rstrip("\n").split("\n")[-1]
    sm = api.create_score_matrix_i32(4, -6)
    for rd in reads[:5]:
        got = api.align_local_gap(rd[10:140], g, sequence_name=("lg", 2), score_matrix=sm, o=-5, e=-1)
        exp = og.align(oracle.M3, rd[10:140], name="lg", idx=2, scores=_table(oracle, sm), o=-5, e=-1)[0]
        assert got == _expect(api, exp)


def test_score_matrix_builders_match_the_oracle(oracle, setup):
    api = setup[0]
    assert _table(oracle, api.create_score_matrix_i32(2, -4)) == oracle.scores_match_mis(2, -4)
    assert _table(oracle, api.create_score_matrix_f32(2, -4)) == oracle.scores_match_mis(2, -4)       # f32 of the i32 matrix
    assert _table(oracle, api._score_matrix_match_mis_f32(2, -4)) == oracle.scores_match_mis(2, -4, f32_variant=True)
    assert _table(oracle, api.create_score_matrix_i32(matrix_file_path=HOXD70)) == oracle.scores_from_mtx(open(HOXD70).read())
    with pytest.raises(Exception):
        api.create_score_matrix_i32(2, None)                           # api.rs:143-146 unwraps both
