"""`-s true` (ambiguous strand) for the POA modes: main.rs:82-106, 132-165, 188-212, 229-253."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def _rc(s):
    return "".join(COMP[c] for c in reversed(s))


def _global_cases():
    """Reads spanning a whole source-to-sink walk (global modes), half of them reverse-complemented."""
    from recgraph_amd import synth
    sg = synth.haplotype_graph(300, 4, path_len=120, seed=11)
    rng = np.random.default_rng(8)
    rd = []
    for k in range(12):
        s = list(sg.path_sequence(k % 4))
        for _ in range(3):
            s[int(rng.integers(0, len(s)))] = "ACGT"[int(rng.integers(0, 4))]
        rd.append("".join(s) if k % 2 == 0 else _rc("".join(s)))
    rd.append("".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=110)))
    return sg.gfa(), ["g%d" % i for i in range(len(rd))], rd


def _cases(example_reads):
    names, reads = example_reads
    rng = np.random.default_rng(7)
    rd = list(reads[:6]) + [_rc(r) for r in reads[6:14]] + [reads[14][:60], _rc(reads[15][20:120])]
    rd.append("".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=90)))     # unrelated on both strands
    rd.append("ACGTNNACGT")
    return ["q%d" % i for i in range(len(rd))], rd


@pytest.mark.parametrize("m,scalar", [(0, False), (0, True), (1, False), (1, True), (2, False), (3, False)])
def test_amb_strand_matches_main_rs(oracle, example_gfa, example_reads, m, scalar):
    from recgraph_amd import api
    if m in (0, 2):
        example_gfa, names, rd = _global_cases()
    else:
        names, rd = _cases(example_reads)
    og = oracle.Graph.from_gfa_text(example_gfa, want_path=False)
    g = api.Graph.from_gfa_text(example_gfa)
    mode = {0: api.MODE_GLOBAL_POA_SCALAR if scalar else api.MODE_GLOBAL_POA,
            1: api.MODE_LOCAL_POA_SCALAR if scalar else api.MODE_LOCAL_POA, 2: api.MODE_GAP_POA,
            3: api.MODE_GAP_LOCAL_POA}[m]
    texts, status = api.align_batch(g, rd, names, mode=mode, amb_strand=True, b=10.0)
    exp = [og.main_rs_amb_strand(m, rd[i], names[i], i + 1, avx2=not scalar, b=10.0) for i in range(len(rd))]
    bad = [(i, texts[i][:300], exp[i][:300]) for i in range(len(rd)) if texts[i] != exp[i]]
    assert not bad, (len(bad), bad[:2])
    assert not any(st & (api.READ_WOULD_PANIC | api.READ_BAD_BASE) for st in status)
    minus = sum("\t-\t" in t for t in texts)
    if m != 3:
        assert minus > 0          # some record came from the reverse-complement run
    else:
        assert minus == 0         # main.rs:240 passes amb_mode = false for -m 3
    # without -s the same reads never carry the reversed labels
    plain, _ = api.align_batch(g, rd, names, mode=mode, b=10.0)
    assert plain != texts
    # the same retry INSIDE the library (rg_stream_opts.amb_strand: what a C / Rust caller and the CLI get): second handle
    # per worker, the comparison and the warning lines in the worker
    for handles, tile in ((2, 5), (1, 100)):
        stexts, sstatus = api.align_stream(g, rd, names, mode=mode, device_ids=[0], handles_per_device=handles, tile_reads=tile,
                                           amb_strand=True, b=10.0)
        assert stexts == exp and sstatus == status, (handles, tile)


def test_amb_strand_option_is_ignored_by_the_pathwise_modes_and_refuses_kept_records():
    from recgraph_amd import api, synth
    sg = synth.haplotype_graph(400, 4, path_len=100, seed=3)
    g = api.Graph.from_gfa_text(sg.gfa())
    rd = synth.haplotype_reads(sg, 9, 100, seed=4, mosaic_frac=0.5)
    base, _ = api.align_batch(g, rd, None, mode=api.MODE_RECOMBINATION)
    got, _ = api.align_stream(g, rd, None, mode=api.MODE_RECOMBINATION, device_ids=[0], tile_reads=4, amb_strand=True)   # main.rs:254-313 ignore -s
    assert got == base
    with pytest.raises(api._lib.RecGraphError):
        api.Stream(g, api.make_params(api.MODE_GLOBAL_POA), device_ids=[0], amb_strand=True, keep_records=True)


def test_amb_strand_cli(oracle, tmp_path, example_gfa, example_reads, capsys):
    from recgraph_amd import cli
    example_gfa, names, rd = _global_cases()
    gp, rp = tmp_path / "g.gfa", tmp_path / "r.fa"
    gp.write_text(example_gfa)
    rp.write_text("".join(">%s\n%s\n" % (names[i], rd[i]) for i in range(len(rd))))
    og = oracle.Graph.from_gfa_text(example_gfa, want_path=False)
    for m in (0, 2):
        cli.main([str(rp), str(gp), "-m", str(m), "-s", "true"])
        out = capsys.readouterr().out
        assert out == "".join(og.main_rs_amb_strand(m, rd[i], names[i], i + 1) for i in range(len(rd)))


def test_rev_and_compl_vectors():
    """sequences.rs:85-100."""
    from recgraph_amd import api
    assert api.rev_and_compl("AAT") == "ATT"
    assert api.rev_and_compl("ATCGN") == "NCGAT"
    with pytest.raises(Exception):
        api.rev_and_compl("AXT")
