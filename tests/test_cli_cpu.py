"""CPU: the CLI front-end's argument surface and FASTA reader follow args_parser.rs / sequences.rs."""
import os


def test_defaults_follow_args_parser_rs():
    from recgraph_amd import cli
    a = cli.build_parser().parse_args(["r.fa", "g.gfa"])
    assert (a.alignment_mode, a.match_score, a.mismatch_score, a.gap_open, a.gap_extension) == (0, 2, 4, 4, 2)
    assert (a.base_rec_cost, a.multi_rec_cost, a.rec_band_width, a.extra_b, a.extra_f) == (4, 0.1, 1.0, 1, 0.01)
    assert a.out_file == "standard output" and a.matrix == "none" and a.amb_strand == "false"
    b = cli.build_parser().parse_args(["r.fa", "g.gfa", "-m", "8", "-R", "7", "-r", "0.25", "-B", "0.5", "-M", "3", "-X", "5"])
    assert (b.alignment_mode, b.base_rec_cost, b.multi_rec_cost, b.rec_band_width, b.match_score, b.mismatch_score) == (8, 7, 0.25, 0.5, 3, 5)


def test_fasta_reader(tmp_path):
    from recgraph_amd import cli
    p = tmp_path / "x.fa"
    p.write_text(">r1 desc\nacg-t\nNN\n\n>r2\nTTTT\n")
    seqs, names = cli.get_sequences(str(p))
    assert names == ["r1 desc", "r2"] and seqs == ["ACGNTNN", "TTTT"]
    here = os.path.dirname(os.path.abspath(__file__))
    seqs, names = cli.get_sequences(os.path.join(here, "golden", "example_reads.fa"))
    assert len(seqs) == len(names) == 52 and all(len(s) == 150 for s in seqs)


def test_fasta_rules_of_sequences_rs():
    """rg_reads_from_fasta against a literal Python restatement of sequences::get_sequences (sequences.rs:5-45) on
    hand-made and random texts: CRLF, blank lines, multi-line records, lower case, '-', names with spaces, a header
    without bases (count mismatch -> refused), bases before the first header (paired with the names by index)."""
    import random
    import pytest
    from recgraph_amd import _lib, api

    def literal(text):
        seqs, names, cur = [], [], []
        lines = text.split("\n")
        for k, line in enumerate(lines):
            if k == len(lines) - 1:
                if line == "":
                    break               # the text ended with '\n' (or is empty): no further line
            elif line.endswith("\r"):
                line = line[:-1]        # BufRead::lines strips "\r\n", not a '\r' at the very end of the file
            if not line.startswith(">") and line != "":
                cur += ["N" if c == "-" else (c.upper() if "a" <= c <= "z" else c) for c in line]
            elif line.startswith(">"):
                names.append(line[1:])
                if cur:
                    seqs.append("".join(cur))
                cur = []
        if cur:
            seqs.append("".join(cur))
        if len(seqs) != len(names):
            raise ValueError("wrong fasta file format")
        return seqs, names

    def lib(text):
        r = api.Reads.from_fasta_text(text)
        assert list(r.offsets) == [sum(len(x) for x in r.sequences()[:i]) for i in range(len(r) + 1)]
        return r.sequences(), r.names

    cases = [">a\nACGT\n", ">a b c\r\nac-gt\r\nNN\r\n\r\n>b\r\nTT", "ACGT\n>a\n>b\nGG\n", ">a\nAC\n\n\nGT\n>b\nT\n", ">\nA\n",
             ">a\nA C\n>b\n \n"]
    for text in cases:
        assert lib(text) == literal(text), text
    for text in (">a\n>b\nAC\n", "ACGT\n", ">a\n", ">a\nAC\n>b\n"):
        with pytest.raises(ValueError):
            literal(text)
        with pytest.raises(_lib.RecGraphError, match="wrong fasta file format"):
            lib(text)
    assert lib("") == ([], []) == literal("")
    rnd = random.Random(7)
    for _ in range(300):
        parts = []
        for _ in range(rnd.randint(0, 8)):
            kind = rnd.random()
            if kind < 0.35:
                parts.append(">" + "".join(rnd.choice("abc 12>") for _ in range(rnd.randint(0, 5))))
            elif kind < 0.9:
                parts.append("".join(rnd.choice("ACGTacgtnN-x") for _ in range(rnd.randint(0, 9))))
            else:
                parts.append("")
        text = "".join(p + rnd.choice(["\n", "\r\n"]) for p in parts)
        if rnd.random() < 0.3 and text.endswith("\n"):
            text = text[:-1]            # (may leave a trailing '\r': it then belongs to the last line)
        try:
            exp = literal(text)
        except ValueError:
            with pytest.raises(_lib.RecGraphError):
                lib(text)
            continue
        assert lib(text) == exp, repr(text)


def test_out_file_semantics_follow_write_gaf(tmp_path):
    """utils.rs:200-219 + the numbers main.rs passes: modes 0-3 create the file at the first read and append after;
    modes 4/5/8/9 pass the 0-based index, so the file is re-created at the SECOND read (main.rs:260,268,311)."""
    from recgraph_amd import cli
    recs = ["rec%d" % i for i in range(4)]
    p = tmp_path / "o.gaf"
    cli.write_gaf_records(str(p), recs, [1, 2, 3, 4])                 # modes 0-3
    assert p.read_text() == "rec0\nrec1\nrec2\nrec3\n"
    cli.write_gaf_records(str(p), recs[:2], [1, 2])                    # an existing file is truncated by number 1
    assert p.read_text() == "rec0\nrec1\n"
    p.unlink()
    cli.write_gaf_records(str(p), recs, [0, 1, 2, 3])                 # modes 4/5/8/9: record 0 is lost
    assert p.read_text() == "rec1\nrec2\nrec3\n"
    cli.write_gaf_records(str(p), recs[:1], [0])                       # one read, file exists: appended
    assert p.read_text() == "rec1\nrec2\nrec3\nrec0\n"
    p.unlink()
    cli.write_gaf_records(str(p), recs[:1], [0])                       # one read, no file: created
    assert p.read_text() == "rec0\n"
    # literal per-read restatement of write_gaf for comparison on random cases
    import os
    import random

    def literal(path, records, numbers):
        for r, n in zip(records, numbers):
            with open(path, "a" if os.path.exists(path) and n != 1 else "w") as f:
                f.write(r + "\n")
    rnd = random.Random(5)
    for case in range(50):
        k = rnd.randint(1, 5)
        base = rnd.choice([0, 1])
        a, b = tmp_path / ("a%d" % case), tmp_path / ("b%d" % case)
        if rnd.random() < 0.5:
            a.write_text("old\n")
            b.write_text("old\n")
        cli.write_gaf_records(str(a), recs[:k], list(range(base, base + k)))
        literal(str(b), recs[:k], list(range(base, base + k)))
        assert a.read_text() == b.read_text()


def test_fasta_check_counts_like_get_sequences(tmp_path):
    """rg_fasta_check (the file-level check the streaming CLI runs ahead of its output: sequences.rs:41-43) fed in
    pieces == the one-piece parser: same read count, same refusals."""
    import random
    import ctypes as C
    import pytest
    from recgraph_amd import _lib, api
    lib = _lib.load()

    def check(text, rnd):
        st = (C.c_int64 * 4)()
        n = C.c_int64(0)
        b = text.encode()
        pos = 0
        while pos < len(b):
            cnt = min(len(b) - pos, rnd.randint(0, 9))
            _lib.check(lib.rg_fasta_check(b[pos:pos + cnt], cnt, 0, st, C.byref(n)))
            pos += cnt
        rc = lib.rg_fasta_check(b"", 0, 1, st, C.byref(n))
        return rc, n.value
    rnd = random.Random(11)
    cases = [">a\nACGT\n", ">a b c\r\nac-gt\r\nNN\r\n\r\n>b\r\nTT", "ACGT\n>a\n>b\nGG\n", ">a\nAC\n\n\nGT\n>b\nT\n", ">\nA\n", "",
             ">a\n>b\nAC\n", "ACGT\n", ">a\n", ">a\nAC\n>b\n", ">a\r\nAC\r", "\r", ">a\n\r", "\r\n>a\r\n\rA\r\n"]
    for _ in range(400):
        parts = []
        for _ in range(rnd.randint(0, 8)):
            kind = rnd.random()
            parts.append(">" + "".join(rnd.choice("abc 12>") for _ in range(rnd.randint(0, 5))) if kind < 0.35 else
                         "".join(rnd.choice("ACGTacgtnN-x\r") for _ in range(rnd.randint(0, 9))) if kind < 0.9 else "")
        text = "".join(p + rnd.choice(["\n", "\r\n"]) for p in parts)
        cases.append(text[:-1] if rnd.random() < 0.3 and text.endswith("\n") else text)
    for text in cases:
        try:
            exp = len(api.Reads.from_fasta_text(text))
        except _lib.RecGraphError:
            exp = None
        rc, n = check(text, rnd)
        assert (rc == 0) == (exp is not None), repr(text)
        if exp is not None:
            assert n == exp, repr(text)
    p = tmp_path / "x.fa"
    p.write_text(">a\nAC\n>b\nGT\n")
    assert api.fasta_check(str(p), block=3) == 2
    p.write_text(">a\nAC\n>b\n")
    with pytest.raises(_lib.RecGraphError, match="wrong fasta file format"):
        api.fasta_check(str(p), block=2)
