"""CPU: what cutting the pathwise DP to a band does to the cells that are kept (tests/c/band_experiment.cpp).

The pathwise recurrence is not a max-plus recurrence per path: members follow their group alpha's directions, so
a kept cell's value depends on alpha decisions arbitrarily far away, and a band changes kept cells in EVERY read.  The
changes die out towards the diagonal (9 columns from the band edge on this test's 300-base / 8-path case, 30 on 400 bases /
12 paths, 53 on the 1 kbp / 32-path reads of config 5: profiles/r04_band_experiment_c5.json), which is why a band can LOOK exact — and why exactness needs a per-row certificate,
not a value test.  This test pins the measured behaviour the design decision rests on."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_a_band_changes_kept_cells_and_the_changes_stop_short_of_the_diagonal(tmp_path):
    from recgraph_amd import synth
    exe = tmp_path / "band_experiment"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tests", "c", "band_experiment.cpp"),
                           os.path.join(ROOT, "oracle", "orc_common.cpp")])
    sg = synth.haplotype_graph(1500, 8, path_len=300, seed=31)
    (tmp_path / "g.gfa").write_text(sg.gfa())
    (tmp_path / "r.txt").write_text("\n".join(synth.haplotype_reads(sg, 12, length=300, seed=32, mosaic_frac=0.5)) + "\n")
    out = subprocess.run([str(exe), str(tmp_path / "g.gfa"), str(tmp_path / "r.txt"), "12", "16", "64", "120"], capture_output=True, text=True,
                         check=True, timeout=600, env=dict(os.environ, BAND_CERT="1")).stdout
    d = json.loads(out)
    by_w = {b["w"]: b for b in d["bands"]}
    # a narrow band corrupts even the cells on the diagonal and the paths' final scores
    assert by_w[16]["reads_with_changed_core_cells"] == d["reads"] and by_w[16]["reads_with_changed_sink_value"] > 0
    for w in (64, 120):
        b = by_w[w]
        assert b["reads_with_changed_kept_cells"] == d["reads"]          # every read: kept cells are NOT the full DP's
        assert b["changed_cells"] > 10000
        assert b["deepest_change_columns_from_edge"] < 48                # ... but the changes stay near the edge on this data
        assert b["reads_with_changed_core_cells"] == 0 and b["reads_with_changed_sink_value"] == 0
        # the certificate that would make the band exact by construction (per path: certified interval + a scalar upper bound
        # of everything beyond it) is SOUND on this data — no certified cell differs from the full DP — and useless: on some
        # row of some read the certified interval does not even reach its path's diagonal
        c = b["certificate"]
        assert c["certified_cells"] > 100000 and c["certified_cells_that_differ_from_the_full_dp"] == 0
        assert min(c["smallest_certified_reach_left_of_a_diagonal"], c["smallest_certified_reach_right_of_a_diagonal"]) < 12
    # the depth does not shrink with a wider band: it is a property of the data (error density, read length), not of w
    assert abs(by_w[64]["deepest_change_columns_from_edge"] - by_w[120]["deepest_change_columns_from_edge"]) <= 8
