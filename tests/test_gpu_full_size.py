"""GPU, BASELINE.json's full configuration sizes: size-independent properties (determinism, shard
invariance, CIGAR/score consistency recomputed independently of both implementations) plus a byte-for-byte
comparison of EVERY read of the sample with the oracle (its threaded runner: ~0.8 s per read per host thread at
config 5)."""
import os
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


SPEC_MARGIN_DEFAULT = 112     # rg_host.hpp (what the tests put back after changing it)

def _parse(line):
    f = line.rstrip("\n").split("\t", 12)
    comments = f[12]
    cigar = comments.split(",")[0]
    ops = [(int(n), c) for n, c in re.findall(r"(\d+)([MXID])", cigar)]
    return f, comments, ops


def _cigar_score(ops, path_bases, read, M=2, X=-4, G=-8):
    """Independent re-scoring of a pathwise GAF record: M/X consume one base of both, I a graph base, D a read base."""
    i = j = 0
    s = 0
    for n, c in ops:
        for _ in range(n):
            if c == "M":
                assert path_bases[i] == read[j]
                s += M; i += 1; j += 1
            elif c == "X":
                assert path_bases[i] != read[j]
                s += X; i += 1; j += 1
            elif c == "I":
                s += G; i += 1
            else:
                s += G; j += 1
    assert i == len(path_bases) and j == len(read)
    return s


@pytest.fixture(scope="module")
def c5():
    from recgraph_amd import api, synth
    sg, reads, _ = synth.make_config("C5", n_reads=384)
    return sg, reads, api.Graph.from_gfa_text(sg.gfa())


def test_c5_m8_properties(oracle, c5):
    from recgraph_amd import api
    sg, reads, g = c5
    names = ["r%d" % i for i in range(len(reads))]
    texts, status = api.align_batch(g, reads, names, mode=api.MODE_RECOMBINATION)
    assert not any(status)
    # determinism + shard invariance: any partition of the batch gives the same records
    t2a, _ = api.align_batch(g, reads[:100], names[:100], mode=api.MODE_RECOMBINATION)
    t2b, _ = api.align_batch(g, reads[100:], names[100:], mode=api.MODE_RECOMBINATION, seq_index_base=101)
    assert t2a + t2b == texts
    nrec = 0
    for rd, t in zip(reads, texts):
        f, comments, ops = _parse(t)
        assert f[1] == "1000" and f[2] == "0" and f[3] == "999" and f[4] == "+"
        if "recombination path" in comments:
            nrec += 1
            m = re.search(r"score: ([-0-9.]+), displacement: (\d+)\t([ACGTN]+)\t(\d+)$", comments)
            pb = m.group(3)
            # the printed score is the shortest decimal of an f32 (Rust `{}`), judged by numpy's independent Dragon4
            assert m.group(1) == np.format_float_positional(np.float32(m.group(1)), unique=True, trim="-")
            s = _cigar_score(ops, pb, rd)
            # The CIGAR must consume exactly the read and the spelled path (checked inside _cigar_score).  Its
            # re-computed score is NOT tied to the reported one: m and w follow their group alpha's directions
            #  while the traceback re-maximises on the path's own layer, and the reverse walker reads
            # the delta-encoded row F of w (A.6) - both are reference behaviour.  Only a loose sanity bound holds.
            pen = np.float32(4) + np.float32(0.1) * np.float32(int(m.group(2)))
            assert abs(float(np.float32(s) - pen) - float(m.group(1))) < 400
            assert 0 <= int(m.group(4)) < len(pb)
        else:
            m = re.search(r"best path: (\d+), score: (-?\d+)\t([ACGTN]+)$", comments)
            assert _cigar_score(ops, m.group(3), rd) >= int(m.group(2))
            assert m.group(3) == sg.path_sequence(int(m.group(1)))      # global alignment spells the whole path
    assert 0 < nrec < len(reads)                  # both GAF shapes occur at config 5
    # all 384 reads byte for byte against the oracle (second restatement, sharded over the host threads)
    og = oracle.Graph.from_gfa_text(sg.gfa())
    _, _, exp = og.bench_text(oracle.M8_ABS, reads, nthreads=min(os.cpu_count() or 1, 96), name_prefix="r", idx_base=1)
    bad = [i for i in range(len(reads)) if texts[i].encode() != exp[i]]
    assert not bad, (len(bad), bad[:5], texts[bad[0]][-200:], exp[bad[0]][-200:])
    # the same reads with a ZERO speculation margin (k_verify sends every read whose optimum is not its picked path's score
    # through the second pass with the provable bound) and with the plain step tables: the same bytes
    # (dsel_edge 1000: no always-stored edge rows — about half of these reads end on another path for their last columns and come
    # back for the second pass, which stores every direction word; no_dsel: every word in the first pass)
    for name, val in (("spec_margin", 0), ("no_split", 1), ("no_retire", 1), ("no_pick2", 1), ("no_order", 1), ("dsel_edge", 1000), ("no_dsel", 1)):
        try:
            api.set_option(name, val)
            again, _ = api.align_batch(g, reads[:64], names[:64], mode=api.MODE_RECOMBINATION)
        finally:
            api.set_option(name, SPEC_MARGIN_DEFAULT if name == "spec_margin" else (8 if name == "dsel_edge" else 0))
        assert again == texts[:64], name


def test_c5_path_retirement_skips_work_and_says_so(c5):
    """Path retirement (k_sweep16) and the two-path pick are exact — the byte comparisons above — and they must actually
    bite at config 5: the `performed` counter (member-row updates the sweeps carried out) falls clearly below its value
    with retirement off, while the `counted` one (the reference's cell updates: the unit of the CUPS figures) does not
    move."""
    from recgraph_amd import api
    sg, reads, g = c5
    got = {}
    for key, opts in (("default", ()), ("no_retire", (("no_retire", 1),)), ("one_path_picks", (("no_pick2", 1),))):
        try:
            for name, val in opts:
                api.set_option(name, val)
            b = api.Batch(g, reads[:256], api.make_params(api.MODE_RECOMBINATION))
            b.run()
            b.fetch()
            got[key] = (b.cell_updates, b.cell_updates_performed)
        finally:
            for name, _ in opts:
                api.set_option(name, 0)
    assert got["default"][0] == got["no_retire"][0] == got["one_path_picks"][0]
    assert got["default"][1] < 0.85 * got["no_retire"][1], got
    assert got["default"][1] <= got["one_path_picks"][1], got
    assert got["no_retire"][1] <= got["no_retire"][0]


def test_c5_huge_recombination_cost_equals_best_single_path(c5):
    from recgraph_amd import api
    sg, reads, g = c5
    names = ["r%d" % i for i in range(64)]
    t8, _ = api.align_batch(g, reads[:64], names, mode=api.MODE_RECOMBINATION, R=100000)
    t4, _ = api.align_batch(g, reads[:64], names, mode=api.MODE_PATHWISE)
    for a, b in zip(t8, t4):
        assert "recombination path" not in a
        sa = int(re.search(r"score: (-?\d+)\t", a).group(1))
        sb = int(re.search(r"score: (-?\d+)\t", b).group(1))
        assert sa == sb                      # same best score; tie-break on the path id differs (m4 highest, m8 lowest)


def test_c4_m4_properties(oracle):
    from recgraph_amd import api, synth
    sg, reads, _ = synth.make_config("C4", n_reads=256)
    g = api.Graph.from_gfa_text(sg.gfa())
    names = ["r%d" % i for i in range(len(reads))]
    texts, status = api.align_batch(g, reads, names, mode=api.MODE_PATHWISE)
    assert not any(status)
    for rd, t in zip(reads, texts):
        f, comments, ops = _parse(t)
        m = re.search(r"best path: (\d+), score: (-?\d+)\t([ACGTN]+)$", comments)
        sc = _cigar_score(ops, m.group(3), rd)
        assert sc >= int(m.group(2))
        if int(m.group(1)) == 0:
            assert sc == int(m.group(2))     # path 0 is always its group's alpha: a plain NW optimum
    og = oracle.Graph.from_gfa_text(sg.gfa())
    _, _, exp = og.bench_text(oracle.M4_ABS, reads, nthreads=min(os.cpu_count() or 1, 96), name_prefix="r", idx_base=1)
    bad = [i for i in range(len(reads)) if texts[i].encode() != exp[i]]
    assert not bad, (len(bad), bad[:5])


@pytest.mark.parametrize("cfg", ["C4", "C5"])
def test_full_launch_every_wave_slot_vs_oracle(oracle, cfg):
    """ONE 4096-read launch (the bench's tile: every CU holds its full complement of sweep waves, twelve at config 4) with
    reads of the whole index range compared with the oracle.  Round 5's `-m 4` variant gave low, run-to-run different scores
    for ~25 % of the reads behind index 768 of such a launch and for none of a 256-read batch: a VALU instruction overwrote
    the data registers of the `buffer_store_dwordx4` in front of it (rg_sweep16.hip, st_row), which bites only when the
    memory pipeline is backed up.  The small batches of the other tests never got there."""
    from recgraph_amd import api, synth
    sg, _, _ = synth.make_config(cfg, n_reads=1)
    g = api.Graph.from_gfa_text(sg.gfa())
    mode, om = (api.MODE_RECOMBINATION, oracle.M8_ABS) if cfg == "C5" else (api.MODE_PATHWISE, oracle.M4_ABS)
    reads = synth.haplotype_reads(sg, 4096, 1000, seed=5683, mosaic_frac=0.5 if cfg == "C5" else 0.0)
    names = ["read%d" % i for i in range(len(reads))]
    check = list(range(0, 4096, 32)) + [4095]
    og = oracle.Graph.from_gfa_text(sg.gfa())
    _, _, exp = og.bench_text(om, [reads[i] for i in check], nthreads=min(os.cpu_count() or 1, 96), name_prefix="x")
    for attempt in range(2):        # (the failure was timing dependent: two launches)
        texts, status = api.align_batch(g, reads, names, mode=mode)
        assert not any(status)
        bad = []
        for k, i in enumerate(check):
            # (bench_text numbers the reads by their position in the subset: everything but the trailing read index)
            e = exp[k].decode().replace("x%d\t" % k, "read%d\t" % i, 1)
            if texts[i].rsplit("\t", 1)[0] != e.rsplit("\t", 1)[0]:
                bad.append(i)
        assert not bad, (cfg, attempt, len(bad), bad[:12])


@pytest.mark.parametrize("case", ["c8_400bp", "c4_200bp", "semi_1kbp", "wide_70_paths", "m4_semi_600bp", "len1500", "m4_len1500", "wide_128_paths", "wide_200_paths"])
def test_full_launches_of_the_other_sweep_variants_vs_oracle(oracle, case):
    """The same for the variants configs 4 and 5 do not reach — 8 and 4 columns per lane (four and six-plus waves per SIMD: more
    row stores in flight per CU than anywhere else), the semiglobal flag, more than 64 paths — each as ONE launch that fills
    every wave slot of the chip at the variant's occupancy, reads of the whole index range against the oracle."""
    from recgraph_amd import api, synth
    mode, om, rows, P, rlen, nreads, mosaic = {
        "c8_400bp": (api.MODE_RECOMBINATION, oracle.M8_ABS, 4000, 16, 400, 8192, 0.5),
        "c4_200bp": (api.MODE_RECOMBINATION, oracle.M8_ABS, 2000, 12, 200, 16384, 0.5),
        "semi_1kbp": (api.MODE_RECOMBINATION_SEMI, oracle.M9_ABS, 10000, 32, 1000, 4096, 0.5),
        "wide_70_paths": (api.MODE_RECOMBINATION, oracle.M8_ABS, 6000, 70, 600, 4096, 0.5),
        "m4_semi_600bp": (api.MODE_PATHWISE_SEMI, oracle.M5_ABS, 6000, 16, 600, 8192, 0.0),
        # 32 columns per lane (reads of 1 024 - 2 047 bases): the record variant with gather runs and register runs of two rows, the
        # -m 4 variant with register runs (round 6); 128 paths: two 64-path pages per row
        "len1500": (api.MODE_RECOMBINATION, oracle.M8_ABS, 15000, 32, 1500, 2048, 0.5),
        "m4_len1500": (api.MODE_PATHWISE, oracle.M4_ABS, 15000, 32, 1500, 2048, 0.0),
        "wide_128_paths": (api.MODE_RECOMBINATION, oracle.M8_ABS, 10000, 128, 1000, 2048, 0.5),
        "wide_200_paths": (api.MODE_RECOMBINATION, oracle.M8_ABS, 6000, 200, 600, 2048, 0.5),
    }[case]
    sg = synth.haplotype_graph(rows, P, path_len=rlen, seed=4242)
    g = api.Graph.from_gfa_text(sg.gfa())
    reads = synth.haplotype_reads(sg, nreads, rlen, seed=777, mosaic_frac=mosaic)
    if "semi" in case:
        reads = [r[: len(r) * 3 // 4] for r in reads]
    names = ["read%d" % i for i in range(len(reads))]
    check = sorted(set(list(range(0, nreads, nreads // (48 if rlen > 1000 or P > 64 else 96))) + [nreads - 1]))
    og = oracle.Graph.from_gfa_text(sg.gfa())
    _, _, exp = og.bench_text(om, [reads[i] for i in check], nthreads=min(os.cpu_count() or 1, 96), name_prefix="x")
    texts, status = api.align_batch(g, reads, names, mode=mode)
    assert not any(status)
    bad = []
    for k, i in enumerate(check):
        e = exp[k].decode().replace("x%d\t" % k, "read%d\t" % i, 1)
        if texts[i].rsplit("\t", 1)[0] != e.rsplit("\t", 1)[0]:
            bad.append(i)
    assert not bad, (case, len(bad), bad[:12])


def test_c2_m0_full_config_vs_oracle(oracle):
    """Config 2 at full size (10 000 reads); the oracle is fast enough to compare a 2 000-read stride."""
    from recgraph_amd import api, synth
    sg, reads, _ = synth.make_config("C2")
    assert len(reads) == 10000
    g = api.Graph.from_gfa_text(sg.gfa())
    og = oracle.Graph.from_gfa_text(sg.gfa(), want_path=False)
    names = ["r%d" % i for i in range(len(reads))]
    texts, status = api.align_batch(g, reads, names, mode=api.MODE_GLOBAL_POA)
    assert len(texts) == 10000
    for i in range(0, 10000, 5):
        assert texts[i] == og.align(oracle.M0_SIMD, reads[i], name=names[i], idx=i + 1)[0]
    again, _ = api.align_batch(g, reads[5000:5100], names[5000:5100], mode=api.MODE_GLOBAL_POA, seq_index_base=5001)
    assert again == texts[5000:5100]


def test_c3_m2_full_config_vs_oracle(oracle):
    """Config 3 at its full 10 000 reads, through the streaming engine (three handles share the HBM for the band
    arenas: ~10 MB per read); a 1 000-read stride is compared byte for byte with the oracle."""
    from recgraph_amd import api, synth
    sg, reads, _ = synth.make_config("C3")
    assert len(reads) == 10000
    g = api.Graph.from_gfa_text(sg.gfa())
    og = oracle.Graph.from_gfa_text(sg.gfa(), want_path=False)
    names = ["r%d" % i for i in range(len(reads))]
    texts, status = api.align_stream(g, reads, names, mode=api.MODE_GAP_POA, device_ids=[0], tile_reads=2500)
    assert len(texts) == 10000
    npanic = 0
    for i in range(0, 10000, 10):
        exp, _, panic, _ = og.align(oracle.M2, reads[i], name=names[i], idx=i + 1)
        if panic:
            npanic += 1
            assert status[i] & api.READ_WOULD_PANIC
        else:
            assert texts[i] == exp, i
    assert npanic < 100
    again, _ = api.align_batch(g, reads[7000:7100], names[7000:7100], mode=api.MODE_GAP_POA, seq_index_base=7001)
    assert again == texts[7000:7100]


def test_cli_example_matches_oracle(oracle, example_gfa, example_reads, tmp_path, capsys):
    import os
    from recgraph_amd import cli
    here = os.path.dirname(os.path.abspath(__file__))
    names, reads = example_reads
    og = oracle.Graph.from_gfa_text(example_gfa)
    for m, om in ((4, oracle.M4_ABS), (0, oracle.M0_SIMD)):
        cli.main([os.path.join(here, "golden", "example_reads.fa"), os.path.join(here, "golden", "example_graph.gfa"), "-m", str(m)])
        out = capsys.readouterr().out
        exp = "".join(og.align(om, rd, name=names[i], idx=i + 1)[0] for i, rd in enumerate(reads))
        assert out == exp


def _host_threads(cap):
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(cap, n))


def test_c5_and_c4_against_the_literal_restatement(oracle, c5):
    """VERDICT r5 missing #4 / next #2a.  Everywhere else at configuration size the comparator is the absolute-form oracle
    (`orc_pathwise_abs.cpp`: rolling rows, direction words, candidate lists — the GPU's own shape).  Here 16 config-5 reads
    (eight of each GAF shape) meet `M8_PRUNED` — the branch-by-branch transliteration of
    pathwise_alignment_recombination.rs:23-883 with its full L x (n+1) x P matrices and an exactly pruned best_alignment scan
    (13-19 s and 2.6 GB per read: eight host threads) — and 64 config-4 reads meet `M4`, the transliteration of
    pathwise_alignment.rs:5-340."""
    from recgraph_amd import api, synth
    sg, reads, g = c5
    names = ["r%d" % i for i in range(len(reads))]
    texts, status = api.align_batch(g, reads, names, mode=api.MODE_RECOMBINATION)
    assert not any(status)
    rec = [i for i, t in enumerate(texts) if "recombination path" in t]
    plain = [i for i, t in enumerate(texts) if "recombination path" not in t]
    assert len(rec) >= 8 and len(plain) >= 8
    # spread over the batch: the first, the last and evenly between
    pick = sorted([rec[k * (len(rec) - 1) // 7] for k in range(8)] + [plain[k * (len(plain) - 1) // 7] for k in range(8)])
    og = oracle.Graph.from_gfa_text(sg.gfa())
    _, _, exp = og.bench_text(oracle.M8_PRUNED, [reads[i] for i in pick], nthreads=_host_threads(8), name_prefix="x")
    for k, i in enumerate(pick):
        e = exp[k].decode().replace("x%d\t" % k, "r%d\t" % i, 1)
        assert texts[i].rsplit("\t", 1)[0] == e.rsplit("\t", 1)[0], (i, texts[i][-200:], e[-200:])
    sg4, reads4, _ = synth.make_config("C4", n_reads=64)
    g4 = api.Graph.from_gfa_text(sg4.gfa())
    names4 = ["r%d" % i for i in range(len(reads4))]
    t4, st4 = api.align_batch(g4, reads4, names4, mode=api.MODE_PATHWISE)
    assert not any(st4)
    og4 = oracle.Graph.from_gfa_text(sg4.gfa())
    _, _, exp4 = og4.bench_text(oracle.M4, reads4, nthreads=_host_threads(16), name_prefix="r", idx_base=1)
    bad = [i for i in range(len(reads4)) if t4[i].encode() != exp4[i]]
    assert not bad, (len(bad), bad[:5])


@pytest.mark.parametrize("case", ["m0_simd", "m0_scalar", "m2", "m2_2k_rows", "m1_simd", "m1_scalar", "m3"])
def test_poa_full_launches_with_real_alignments(oracle, case):
    """VERDICT r5 weak #3 / next #2b.  Config 2's own reads are 100 % "band not enough": its full-size test compares warning
    lines and empty records, and the walkers (a13) met the oracle only in batches of a few hundred reads — the size that hid
    round 5's store hazard in `-m 4`.  Here every POA mode runs ONE full-occupancy launch of whole source->sink walks
    (`synth.full_walk_reads`: reads a global alignment places inside the band, so the device walker and the formatter produce
    real CIGARs) and a stride over the whole index range is compared byte for byte with the oracle."""
    from recgraph_amd import api, synth
    mode, om, cfg, nreads, ncheck = {
        "m0_simd": (api.MODE_GLOBAL_POA, oracle.M0_SIMD, "C2", 10000, 500),
        "m0_scalar": (api.MODE_GLOBAL_POA_SCALAR, oracle.M0_SCALAR, "C2", 10000, 500),
        "m2": (api.MODE_GAP_POA, oracle.M2, "C2", 10000, 500),
        "m2_2k_rows": (api.MODE_GAP_POA, oracle.M2, "C3", 8192, 256),
        "m1_simd": (api.MODE_LOCAL_POA, oracle.M1_SIMD, "C2", 8192, 256),
        "m1_scalar": (api.MODE_LOCAL_POA_SCALAR, oracle.M1_SCALAR, "C2", 8192, 256),
        "m3": (api.MODE_GAP_LOCAL_POA, oracle.M3, "C2", 8192, 192),
    }[case]
    sg, _, _ = synth.make_config(cfg, n_reads=1)
    g = api.Graph.from_gfa_text(sg.gfa())
    og = oracle.Graph.from_gfa_text(sg.gfa(), want_path=False)
    reads = synth.full_walk_reads(sg, nreads, seed=20261)
    names = ["w%d" % i for i in range(nreads)]
    texts, status = api.align_batch(g, reads, names, mode=mode)
    assert len(texts) == nreads
    check = sorted(set(list(range(0, nreads, max(1, nreads // ncheck))) + [nreads - 1]))
    walked = 0
    for i in check:
        exp, _, panic, _ = og.align(om, reads[i], name=names[i], idx=i + 1)
        if panic:
            assert status[i] & api.READ_WOULD_PANIC, i
            continue
        assert texts[i] == exp, (case, i, texts[i][-160:], exp[-160:])
        # a real alignment: a GAF line with a CIGAR, not the reference's "band not enough" record
        walked += 1 if "band not enough" not in texts[i] and re.search(r"\t(\d+[MXID])+", texts[i]) else 0
    assert walked > 0.9 * len(check), (case, walked, len(check))


@pytest.mark.parametrize("stripe_c", [8, 16, 32])
def test_full_launch_of_striped_long_reads_vs_oracle(oracle, stripe_c):
    """ADVICE r5 (medium).  The column-striped i32 sweep (reads of 2 048 bases and more: one wave per stripe, FIFO pipeline) met
    the oracle only in batches of a handful of reads, and its 32-columns-per-lane instantiation once returned wrong sink
    values that only its `kOld` control flow avoids (not a hardware hazard — s_nop padding and forced waitcnts
    do not cure it — but a miscompile suspect in a kernel with ~700 spilled registers).  Here each stripe width runs ONE launch
    that gives every CU its workgroups — 512 reads of 2 600 bases, -m 8 and -m 4 — and a stride over the whole index range is
    compared byte for byte with the oracle."""
    from recgraph_amd import api, synth
    sg = synth.haplotype_graph(6500, 4, path_len=2600, seed=2600)
    g = api.Graph.from_gfa_text(sg.gfa())
    nreads = 512
    reads = synth.haplotype_reads(sg, nreads, 2600, seed=4211, mosaic_frac=0.5)
    names = ["read%d" % i for i in range(nreads)]
    check = sorted(set(list(range(0, nreads, 16)) + [nreads - 1]))
    og = oracle.Graph.from_gfa_text(sg.gfa())
    try:
        api.set_option("stripe_c", stripe_c if stripe_c != 16 else 0)
        for mode, om in ((api.MODE_RECOMBINATION, oracle.M8_ABS), (api.MODE_PATHWISE, oracle.M4_ABS)):
            _, _, exp = og.bench_text(om, [reads[i] for i in check], nthreads=_host_threads(16), name_prefix="x")
            texts, status = api.align_batch(g, reads, names, mode=mode)
            assert not any(status)
            bad = []
            for k, i in enumerate(check):
                e = exp[k].decode().replace("x%d\t" % k, "read%d\t" % i, 1)
                if texts[i].rsplit("\t", 1)[0] != e.rsplit("\t", 1)[0]:
                    bad.append(i)
            assert not bad, (stripe_c, mode, len(bad), bad[:12])
    finally:
        api.set_option("stripe_c", 0)
