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
