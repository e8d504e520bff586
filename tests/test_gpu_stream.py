"""GPU: the streaming engine behind the C ABI (rg_stream_*, rg_reads_from_fasta) — the pipeline bench.py times and the
CLI runs — delivers, tile by tile and in input order, exactly the bytes of the plain one-handle batch path (which the
other GPU tests hold equal to the oracle)."""
import os
import subprocess
import threading
import time
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _graph_reads(n=61, seed=21):
    from recgraph_amd import api, synth
    sg = synth.haplotype_graph(1500, 8, path_len=300, seed=seed)
    reads = synth.haplotype_reads(sg, n, length=300, seed=seed + 1, mosaic_frac=0.5)
    return sg, api.Graph.from_gfa_text(sg.gfa()), reads


def test_stream_equals_batch_in_every_mode(oracle):
    from recgraph_amd import api
    sg, g, reads = _graph_reads()
    names = ["q%d" % i for i in range(len(reads))]
    og = oracle.Graph.from_gfa_text(sg.gfa())
    for mode, om in ((api.MODE_RECOMBINATION, oracle.M8_ABS), (api.MODE_PATHWISE, oracle.M4_ABS), (api.MODE_GLOBAL_POA, oracle.M0_SIMD),
                     (api.MODE_GAP_POA, oracle.M2), (api.MODE_RECOMBINATION_SEMI, None), (api.MODE_LOCAL_POA, None)):
        one, st1 = api.align_batch(g, reads, names, mode=mode)
        if om is not None:
            assert one[7] == og.align(om, reads[7], name=names[7], idx=8)[0]
        for handles, tile in ((3, 8), (1, 16), (2, 1000)):
            texts, status = api.align_stream(g, reads, names, mode=mode, device_ids=[0], handles_per_device=handles, tile_reads=tile)
            assert texts == one and status == st1, (mode, handles, tile)
    # default names are global: read<i>
    texts, _ = api.align_stream(g, reads, None, mode=api.MODE_PATHWISE, device_ids=[0], tile_reads=9)
    base, _ = api.align_batch(g, reads, None, mode=api.MODE_PATHWISE)
    assert texts == base


def test_stream_pushes_interleaved_with_results_and_statistics():
    from recgraph_amd import api
    _, g, reads = _graph_reads(n=90, seed=31)
    one, _ = api.align_batch(g, reads, None, mode=api.MODE_RECOMBINATION)
    st = api.Stream(g, api.make_params(api.MODE_RECOMBINATION), device_ids=[0, 0], handles_per_device=2, tile_reads=7)
    got, firsts = [], []
    st.push(reads[:30])
    t = st.next()
    got.append(t)
    st.push(api.Batch.pack_reads(reads[30:75]))          # the packed (bytes, offsets) form
    st.push(reads[75:])
    st.finish()
    got += list(st)
    assert st.next() is None                               # RG_STREAM_END is sticky
    for t in got:
        firsts.append(t.first)
        assert t.n <= 7 and t.device == 0 and t.cell_updates > 0
        assert [t.text_of(i).decode() for i in range(t.n)] == one[t.first:t.first + t.n]
        assert t.text == "".join(one[t.first:t.first + t.n]).encode()
    assert firsts == sorted(firsts) and sum(t.n for t in got) == len(reads)
    ks = st.kernel_stats()
    assert any(k.startswith("k_sweep") for k in ks) and "host:format" in ks and 1 <= st.handles <= 4
    nt = len(got)
    assert all(v[1] == nt for k, v in ks.items() if k in ("host:set_reads", "host:run", "host:fetch", "host:format"))
    st.close()
    with pytest.raises(api._lib.RecGraphError):
        api.Stream(g, api.make_params(api.MODE_PATHWISE), device_ids=[99])


def test_stream_reports_bad_reads_and_keeps_going():
    """A read with a character outside ACGTN is flagged (the reference panics on it) and the rest of its tile is aligned;
    an empty read is refused at the push; a tile that fails on the device (reads too long for the pathwise kernels) comes
    back as an error from rg_stream_next and the stream moves on."""
    from recgraph_amd import api
    _, g, reads = _graph_reads(n=12, seed=41)
    rd = list(reads)
    rd[5] = rd[5][:50] + "X" + rd[5][51:]
    st = api.Stream(g, api.make_params(api.MODE_PATHWISE), device_ids=[0], tile_reads=4)
    with pytest.raises(api._lib.RecGraphError):
        st.push(rd[:3] + [""])
    st.push(rd)
    st.push(["ACGT" * 4200])            # 16 800 bases: RG_ERR_ARG from the batch run
    st.push(rd[:2])
    st.finish()
    one, _ = api.align_batch(g, reads, None, mode=api.MODE_PATHWISE)
    seen, errors = 0, 0
    while True:
        try:
            t = st.next()
        except api._lib.RecGraphError as ex:
            errors += 1
            assert "16383" in str(ex)
            continue
        if t is None:
            break
        for i in range(t.n):
            k = t.first + i
            if k == 5:
                assert t.status[i] & api.READ_BAD_BASE and t.text_of(i) == b""
            elif k < 12:
                assert t.text_of(i).decode() == one[k]
            seen += 1
    assert errors == 1 and seen == 14


def test_fasta_in_gaf_out_through_the_library_and_the_cli(tmp_path, oracle, example_gfa):
    from recgraph_amd import api
    fa = os.path.join(ROOT, "tests", "golden", "example_reads.fa")
    gfa = os.path.join(ROOT, "tests", "golden", "example_graph.gfa")
    rd = api.Reads.from_fasta(fa)
    seqs, names = rd.sequences(), rd.names
    og = oracle.Graph.from_gfa_text(example_gfa)
    g = api.Graph.from_gfa_text(example_gfa)
    for mode, om in ((8, oracle.M8_ABS), (0, oracle.M0_SIMD), (4, oracle.M4_ABS)):
        exp = "".join(og.align(om, s, name=names[i], idx=i + 1)[0] for i, s in enumerate(seqs))
        st = api.Stream(g, api.make_params(mode), device_ids=[0], tile_reads=20)
        st.push(rd)
        st.finish()
        assert b"".join(t.text for t in st).decode() == exp
        r = subprocess.run([sys.executable, "-m", "recgraph_amd.cli", fa, gfa, "-m", str(mode), "--tile", "16"], capture_output=True,
                           text=True, cwd=ROOT, timeout=600)
        assert r.returncode == 0 and r.stdout == exp and r.stderr.startswith("Done in"), r.stderr[-500:]
    # -o: modes 4+ lose record 0 (utils.rs:200-219 with the 0-based numbers of main.rs:260)
    out = tmp_path / "o.gaf"
    r = subprocess.run([sys.executable, "-m", "recgraph_amd.cli", fa, gfa, "-m", "4", "-o", str(out)], capture_output=True, text=True,
                       cwd=ROOT, timeout=600)
    assert r.returncode == 0 and r.stdout == ""
    exp = [og.align(oracle.M4_ABS, s, name=names[i], idx=i + 1)[0] for i, s in enumerate(seqs)]
    assert out.read_text() == "".join(exp[1:])


def test_push_fasta_pushes_tiles_while_parsing_and_refuses_a_bad_file(example_gfa):
    from recgraph_amd import api
    g = api.Graph.from_gfa_text(example_gfa)
    fa = open(os.path.join(ROOT, "tests", "golden", "example_reads.fa"), "rb").read()
    whole = api.Reads.from_fasta_text(fa)
    base, _ = api.align_batch(g, whole.sequences(), whole.names, mode=api.MODE_GLOBAL_POA)
    st = api.Stream(g, api.make_params(api.MODE_GLOBAL_POA), device_ids=[0], tile_reads=7)
    assert st.push_fasta(fa) == len(whole)
    assert st.push_fasta(b">x y\nacgtn-ACGT\n") == 1          # canonicalised inside: ACGTNNACGT
    st.finish()
    tiles = list(st)
    assert [t.n for t in tiles] == [7] * 7 + [3, 1]
    assert b"".join(t.text for t in tiles[:-1]).decode() == "".join(base)
    one, _ = api.align_batch(g, ["ACGTNNACGT"], ["x y"], mode=api.MODE_GLOBAL_POA, seq_index_base=len(whole) + 1)
    assert tiles[-1].text.decode() == one[0]
    st.close()
    # name / sequence counts differ at the end of the text: refused (sequences.rs:41-43), after the complete reads before it were pushed
    st = api.Stream(g, api.make_params(api.MODE_GLOBAL_POA), device_ids=[0], tile_reads=2)
    with pytest.raises(api._lib.RecGraphError, match="wrong fasta file format"):
        st.push_fasta(b">a\nACGT\n>b\nGGCC\n>c\nTTAA\n>d\n")
    st.close()


def test_two_distinct_devices():
    """The shared tile queue over two different GPUs (skipped on a one-GPU box)."""
    from recgraph_amd import _lib, api
    if _lib.load().rg_device_count() < 2:
        pytest.skip("one GPU visible")
    _, g, reads = _graph_reads(n=64, seed=51)
    one, _ = api.align_batch(g, reads, None, mode=api.MODE_RECOMBINATION)
    st = api.Stream(g, api.make_params(api.MODE_RECOMBINATION), device_ids=[0, 1], handles_per_device=2, tile_reads=4)
    st.push(reads)
    st.finish()
    tiles = list(st)
    assert b"".join(t.text for t in tiles).decode() == "".join(one)
    assert {t.device for t in tiles} == {0, 1}
    m = api.MultiBatch(g, reads, api.make_params(api.MODE_PATHWISE), device_ids=[1, 0])
    base, _ = api.align_batch(g, reads, None, mode=api.MODE_PATHWISE)
    assert m.format_all(["read%d" % i for i in range(len(reads))], 1, 4).decode() == "".join(base)


def test_a_failed_set_reads_leaves_no_stale_handle():
    """a handle whose rg_batch_set_reads was refused keeps its previous read set; run/fetch never see a
    half-loaded one."""
    from recgraph_amd import api
    _, g, reads = _graph_reads(n=10, seed=61)
    b = api.Batch(g, reads, api.make_params(api.MODE_PATHWISE))
    b.run()
    b.fetch()
    before = [b.gaf_text(i, "r", i + 1) for i in range(10)]
    with pytest.raises(api._lib.RecGraphError):
        b.set_reads(reads[:3] + [""] + reads[3:] * 40)         # refused before anything of the handle is touched
    b.n = 10
    b.run()
    b.fetch()
    assert [b.gaf_text(i, "r", i + 1) for i in range(10)] == before


def test_bounded_stream_with_a_slow_consumer():
    """max_queued_tiles / max_undelivered_bytes bound what the stream holds: with a cap of 2 queued tiles and
    ~2 tiles of finished text, a consumer that sleeps never sees more than a handful of tiles pending while a feeder thread
    pushes 40 of them; the bytes are those of the unbounded run."""
    import threading
    import time
    from recgraph_amd import api
    _, g, reads = _graph_reads(n=200, seed=71)
    one, _ = api.align_batch(g, reads, None, mode=api.MODE_PATHWISE)
    tile_bytes = len("".join(one[:5]))
    st = api.Stream(g, api.make_params(api.MODE_PATHWISE), device_ids=[0], handles_per_device=2, tile_reads=5,
                    max_queued_tiles=2, max_undelivered_bytes=2 * tile_bytes)
    peak = [0]
    done = threading.Event()

    def feeder():
        for k in range(0, len(reads), 5):
            st.push(reads[k:k + 5])                  # blocks while 2 tiles are queued
            peak[0] = max(peak[0], st.pending)
        st.finish()
        done.set()
    th = threading.Thread(target=feeder)
    th.start()
    got = []
    for t in st:
        got.append(t)
        if len(got) < 12:
            time.sleep(0.05)                         # the slow consumer
            assert not done.is_set()                 # the feeder is held back: 40 tiles cannot all be in
            assert st.pending <= 10                  # queued (2) + on the handles (2) + finished (the 2-tile cap is crossed by what was in flight)
    th.join()
    assert peak[0] <= 10
    assert b"".join(t.text for t in got).decode() == "".join(one) and len(got) == 40
    st.close()


def test_abort_wakes_a_blocked_feeder_and_destroy_is_safe(example_gfa, example_reads):
    """a bounded stream whose consumer gives up.  The feeder sits inside a push that waits for room in the queue
    (nobody drains); `rg_stream_abort` makes that push — and every later call — return an error at once, `next` fails the same
    way, and the stream can be closed while tiles are still on the device.  A failed `feed_fasta` leaves no stale carry."""
    from recgraph_amd import api
    names, reads = example_reads
    g = api.Graph.from_gfa_text(example_gfa)
    st = api.Stream(g, api.make_params(api.MODE_PATHWISE), device_ids=[0], handles_per_device=1, tile_reads=4, max_queued_tiles=1)
    pushed, err = [0], []
    started = threading.Event()

    def feeder():
        try:
            for k in range(0, 400):
                st.push(reads[(4 * k) % 48:(4 * k) % 48 + 4])         # blocks once one tile is queued and one is on the handle
                pushed[0] += 1
                started.set()
        except Exception as ex:
            err.append(ex)
    th = threading.Thread(target=feeder)
    th.start()
    assert started.wait(60)
    time.sleep(0.3)                                  # the feeder is now inside a blocked push
    n_before = pushed[0]
    assert th.is_alive() and n_before < 400
    st.abort()
    th.join(30)
    assert not th.is_alive() and len(err) == 1 and "abort" in str(err[0])
    with pytest.raises(Exception):
        st.next()
    with pytest.raises(Exception):
        st.push(reads[:4])
    st.close()
    # a feed that fails in the middle (an empty read) resets the feeder: the same stream takes a new text from scratch
    st = api.Stream(g, api.make_params(api.MODE_PATHWISE), device_ids=[0], tile_reads=3)
    good = b"".join(b">%s\n%s\n" % (names[i].encode(), reads[i].encode()) for i in range(5))
    with pytest.raises(Exception):
        st.feed_fasta(good + b">empty\n\n>x\nACGT\n", final=True)
    n = st.feed_fasta(good, final=True)
    st.finish()
    tiles = list(st)
    texts = b"".join(t.text for t in tiles).decode()
    assert texts.count("\n") >= 5 and all(nm.split()[0] in texts for nm in names[:5])
    st.close()


def test_feed_fasta_in_pieces_and_pathwise_read_zero(example_gfa):
    """rg_stream_feed_fasta: any split of the text gives the reads of the one-piece call; a pathwise stream with
    seq_index_base = 0 (what main.rs:260,268,311 pass) still prints read 0 (only the POA modes read the index as
    "score only")."""
    from recgraph_amd import api
    g = api.Graph.from_gfa_text(example_gfa)
    fa = open(os.path.join(ROOT, "tests", "golden", "example_reads.fa"), "rb").read()
    whole = api.Reads.from_fasta_text(fa)
    for mode in (api.MODE_PATHWISE, api.MODE_GLOBAL_POA):
        base, _ = api.align_batch(g, whole.sequences(), whole.names, mode=mode, seq_index_base=0)
        assert ("\t" not in base[0]) == (mode == api.MODE_GLOBAL_POA) and "\t" in base[1]      # "score only" is a POA notion
        for piece in (1 << 20, 777, 13):
            st = api.Stream(g, api.make_params(mode), device_ids=[0], tile_reads=9, seq_index_base=0)
            n = 0
            for k in range(0, len(fa), piece):
                n += st.feed_fasta(fa[k:k + piece])
            n += st.feed_fasta(b"", final=True)
            st.finish()
            assert n == len(whole)
            assert b"".join(t.text for t in st).decode() == "".join(base), (mode, piece)
            st.close()
    # a second text on the same stream after the first was closed, and a refused one
    st = api.Stream(g, api.make_params(api.MODE_GLOBAL_POA), device_ids=[0], tile_reads=4)
    assert st.feed_fasta(b">a\nAC", final=False) == 0 and st.feed_fasta(b"GT\n>b\nGG\n", final=True) == 2
    with pytest.raises(api._lib.RecGraphError, match="wrong fasta file format"):
        st.feed_fasta(b">c\nAA\n>d\n", final=True)
    st.finish()
    assert sum(t.n for t in st) == 3          # the complete read before the bad end was pushed
    st.close()


def test_kept_records_can_be_released():
    from recgraph_amd import api
    _, g, reads = _graph_reads(n=30, seed=81)
    one, _ = api.align_batch(g, reads, None, mode=api.MODE_PATHWISE)
    st = api.Stream(g, api.make_params(api.MODE_PATHWISE), device_ids=[0], tile_reads=7, keep_records=True)
    st.push(reads)
    st.finish()
    for t in st:
        view = api._ShardView(C_void(t.records), t.n, st)
        assert [view.gaf_text(i, "read%d" % (t.first + i), t.first + i + 1) for i in range(t.n)] == one[t.first:t.first + t.n]
        st.release(t)                              # the handle is freed now, not at rg_stream_destroy
        st.release(t)                              # (idempotent)
    st.close()


def C_void(p):
    import ctypes
    return ctypes.c_void_p(p)


def test_cli_refuses_a_malformed_file_before_any_output(tmp_path, example_gfa):
    """The reference parses the whole FASTA file before it aligns (and panics: sequences.rs:41-43): the streaming CLI prints
    nothing for such a file, however many complete reads precede the bad end."""
    gfa = os.path.join(ROOT, "tests", "golden", "example_graph.gfa")
    fa = open(os.path.join(ROOT, "tests", "golden", "example_reads.fa")).read()
    bad = tmp_path / "bad.fa"
    bad.write_text(fa + ">tail without bases\n")
    r = subprocess.run([sys.executable, "-m", "recgraph_amd.cli", str(bad), gfa, "-m", "0", "--tile", "8"], capture_output=True, text=True,
                       cwd=ROOT, timeout=600)
    assert r.returncode != 0 and r.stdout == "" and "wrong fasta file format" in r.stderr
