"""CPU: GFA shapes the reference accepts beyond the tidy synthetic graphs: several sinks, segment ids
that are neither contiguous nor listed in order, L lines in arbitrary order, a segment only one path visits.

Where the reference iterates a HashMap (the sinks that become predecessors of F, graph.rs:112-123; PredHash,
pathwise_graph.rs:75-93) its order is per-process random; the product's documented rule is ASCENDING ROW
ORDER, asserted here.  Everything else is determined by the reference: handles sorted by id (graph.rs:32-33), predecessors
of a segment in the order of the L lines that target it (graph.rs:75)."""
import random

import pytest


@pytest.fixture(scope="module")
def rg():
    from recgraph_amd import _lib
    _lib.build_library()
    import recgraph_amd
    return recgraph_amd


S = {10: "AC", 3: "G", 7: "T", 20: "CA", 15: "TT"}
LINKS = [(7, 20), (7, 15), (3, 10), (10, 15), (3, 7)]
PATHS = [("p0", [3, 10, 15]), ("p1", [3, 7, 20]), ("p2", [3, 7, 15])]


def odd_gfa(links=LINKS, seg_order=(10, 3, 7, 20, 15)):
    out = ["H\tVN:Z:1.0"]
    out += ["S\t%d\t%s" % (i, S[i]) for i in seg_order]
    out += ["L\t%d\t+\t%d\t+\t0M" % l for l in links]
    out += ["P\t%s\t%s\t*" % (n, ",".join("%d+" % i for i in st)) for n, st in PATHS]
    return "\n".join(out) + "\n"


def test_hand_derived_arrays(rg, oracle):
    from recgraph_amd import api
    g = api.Graph.from_gfa_text(odd_gfa())
    # handles sorted by id 3, 7, 10, 15, 20 -> rows 1 | 2 | 3-4 | 5-6 | 7-8, F = 9
    assert g.dump(0) == "$GTACTTCAF"
    assert g.dump(15) == "0,3,7,10,10,15,15,20,20,0"
    # sources get predecessor 0 (graph.rs:64-74); row 5 lists 7 (row 2) before 10 (row 4): L-line order, not row order;
    # F lists the two sinks 15 (row 6) and 20 (row 8) in ascending row order (the documented tie rule)
    assert g.dump(2) == "1:0;2:1;3:1;5:2,4;7:2;9:6,8;"
    assert g.dump(1) == "0111010101"
    # PathGraph: predecessors come from consecutive path steps (pathwise_graph.rs:207-233), ascending; segment 20 (rows 7-8) is on p1 only
    assert g.paths_number == 3
    assert g.dump(12) == "1:0=111;2:1=011;3:1=100;5:2=001,4=100;7:2=010;9:6=101,8=010;"
    assert g.dump(13).split(";")[7] == "010" and g.dump(14) == "0,0,1,0,0,0,0,1,1,0"
    og = oracle.Graph.from_gfa_text(odd_gfa())
    for which in (0, 1, 2, 3, 4, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19):
        assert g.dump(which) == og.dump(which), which


def test_line_order_rules(rg, oracle):
    """S lines in any order give the same graph; the order of the L lines only permutes the predecessors inside a row."""
    from recgraph_amd import api
    rnd = random.Random(3)
    ref = api.Graph.from_gfa_text(odd_gfa())
    for _ in range(10):
        so = list(S)
        rnd.shuffle(so)
        g = api.Graph.from_gfa_text(odd_gfa(seg_order=so))
        assert [g.dump(w) for w in (0, 2, 12, 15)] == [ref.dump(w) for w in (0, 2, 12, 15)]
        lk = LINKS[:]
        rnd.shuffle(lk)
        g, og = api.Graph.from_gfa_text(odd_gfa(links=lk)), oracle.Graph.from_gfa_text(odd_gfa(links=lk))
        want5 = [2 if a == 7 else 4 for a, b in lk if b == 15]
        assert g.dump(2) == "1:0;2:1;3:1;5:%d,%d;7:2;9:6,8;" % tuple(want5)
        assert g.dump(2) == og.dump(2) and g.dump(12) == ref.dump(12)       # PredHash does not depend on L lines at all


def renumbered(sg, rnd):
    """A synthetic graph with ids mapped to a random increasing sequence (gaps, still topological) and shuffled lines."""
    ids = sorted(rnd.sample(range(1, 5 * len(sg.segments)), len(sg.segments)))
    mp = {i: ids[k] for k, (i, _) in enumerate(sg.segments)}
    segs = ["S\t%d\t%s" % (mp[i], s) for i, s in sg.segments]
    links = ["L\t%d\t+\t%d\t+\t0M" % (mp[a], mp[b]) for a, b in sorted(set(sg.links))]
    rnd.shuffle(segs)
    rnd.shuffle(links)
    paths = ["P\tp%d\t%s\t*" % (k, ",".join("%d+" % mp[i] for i in p)) for k, p in enumerate(sg.paths)]
    lines = segs + links
    rnd.shuffle(lines)                     # S and L lines interleaved
    return "\n".join(["H\tVN:Z:1.0"] + lines + paths) + "\n"


def test_renumbered_shuffled_graphs_match_the_oracle(rg, oracle):
    from recgraph_amd import api, synth
    rnd = random.Random(11)
    for sg in (synth.haplotype_graph(500, 6, path_len=100, seed=2), synth.linear_graph(300, seed=5)):
        for _ in range(3):
            t = renumbered(sg, rnd)
            g, og = api.Graph.from_gfa_text(t), oracle.Graph.from_gfa_text(t)
            for which in (0, 1, 2, 3, 4, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19):
                assert g.dump(which) == og.dump(which), which
            assert g.dump(0) == api.Graph.from_gfa_text(sg.gfa()).dump(0)     # same linearisation as the tidy numbering


def test_random_walk_graphs_are_well_formed():
    """`synth.random_dag_graph` (the non-block topologies of the GPU parity tests and the fuzz campaign): ids 1..S in
    topological order, every segment on a path, every path from segment 1 to segment S in ascending id order, links =
    the consecutive pairs of the paths; the library takes the GFA with all its paths."""
    from recgraph_amd import api, synth
    for seed, nseg, P, kw in ((1, 30, 4, {}), (2, 80, 17, {"max_jump": 6}), (3, 12, 64, {"max_seg": 2}), (4, 50, 1, {})):
        g = synth.random_dag_graph(nseg, P, seed=seed, **kw)
        assert [i for i, _ in g.segments] == list(range(1, nseg + 1))
        assert all(len(sq) >= 1 and set(sq) <= set("ACGT") for _, sq in g.segments)
        used = set()
        for p in g.paths:
            assert p[0] == 1 and p[-1] == nseg and all(a < b for a, b in zip(p, p[1:]))
            used.update(p)
        assert used == set(range(1, nseg + 1))
        assert set(g.links) == {(a, b) for p in g.paths for a, b in zip(p, p[1:])}
        gg = api.Graph.from_gfa_text(g.gfa())
        assert gg.paths_number == P and not gg.path_error
        assert gg.rows == g.rows
