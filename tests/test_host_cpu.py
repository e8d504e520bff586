"""CPU: host-side logic of the product (no GPU compute): the C-ABI library loads and exports every
symbol include/recgraph_hip.h declares, graph flattening equals the oracle's arrays, score matrices and
defaults follow the reference, and batch calls fail loudly without a HIP device."""
import ctypes as C
import json
import os
import re

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
VEC = json.load(open(os.path.join(HERE, "golden", "reference_unit_vectors.json")))


@pytest.fixture(scope="module")
def rg():
    from recgraph_amd import _lib
    _lib.build_library()
    import recgraph_amd
    return recgraph_amd


def test_library_exports_every_declared_symbol(rg):
    from recgraph_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "recgraph_hip.h")).read()
    declared = set(re.findall(r"\b(rg_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    lib = C.CDLL(_lib.library_path())
    for sym in sorted(declared):
        assert hasattr(lib, sym), sym
    assert declared == set(_lib.SYMBOLS)
    # ... and NOTHING else: a drop-in library exports its ABI, not its kernels' host stubs, launchers or C++ helpers
    # (csrc/exports.map)
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.library_path()], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert exported == declared, sorted(exported ^ declared)


def test_flattening_matches_oracle(rg, oracle, example_gfa):
    from recgraph_amd import api, synth
    texts = [example_gfa, synth.linear_graph(400, seed=3).gfa(), synth.haplotype_graph(900, 7, path_len=150, seed=4).gfa()]
    texts += [v["gfa"] for v in VEC["path_graph"]]
    for t in texts:
        g, og = api.Graph.from_gfa_text(t), oracle.Graph.from_gfa_text(t)
        for which in (0, 1, 2, 3, 4, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19):
            assert g.dump(which) == og.dump(which), which


@pytest.mark.parametrize("v", VEC["graph_struct"] + VEC["path_graph"], ids=lambda v: v["ref"])
def test_reference_graph_unit_vectors(rg, v):
    from recgraph_amd import api
    g = api.Graph.from_gfa_text(v["gfa"])
    if "lnz" in v:
        assert g.dump(0) == v["lnz"]
    if "paths_number" in v:
        assert g.paths_number == v["paths_number"]
        rows = g.dump(13).strip(";").split(";")
        for row, bits in v["paths_nodes"].items():
            assert rows[int(row)] == bits
    if "preds" in v:
        preds = dict(x.split(":") for x in g.dump(2).strip(";").split(";"))
        for k, p in v["preds"].items():
            assert [int(t) for t in preds[k].split(",")] == p


def test_score_matrices_and_defaults(rg):
    from recgraph_amd import api
    m = api.create_score_matrix_i32(10, -10)
    assert m[("A", "A")] == 10 and m[("A", "C")] == -10 and m[("N", "N")] == -10 and ("-", "-") not in m
    assert m[("A", "-")] == -20                     # any pairing with '-' = 2x (score_matrix.rs:43)
    f = api.create_score_matrix_f32(2, -4)
    assert f[("A", "-")] == -8.0 and f[("A", "A")] == 2.0          # api.rs:153-164: the i32 matrix as f32 (gap = 2x)
    d = api._score_matrix_match_mis_f32(2, -4)                     # default of align_*_no_gap (score_matrix.rs:52-66)
    assert d[("A", "-")] == -4.0 and d[("N", "N")] == -4.0 and ("-", "-") not in d
    with pytest.raises(Exception):
        api.create_score_matrix_i32(2, None)                       # api.rs:143-146 unwraps both
    h = api.create_score_matrix_i32(matrix_file_path=os.path.join(HERE, "golden", "HOXD70.mtx"))
    assert h[("A", "A")] == 91 and h[("T", "G")] == -144 and h[("G", "T")] == -114 and h[("A", "-")] == -200
    p = api.make_params(8)
    assert (p.gap_open, p.gap_ext, p.base_rec_cost) == (-4, -2, 4)
    assert abs(p.multi_rec_cost - 0.1) < 1e-7 and p.rec_band_width == 1.0 and p.band_b == 1.0


def test_gaf_struct_roundtrip(rg):
    from recgraph_amd import api
    line = "name\t5\t0\t4\t+\t>1>2>4>6>7\t5\t0\t4\t0\t*\t*\t5M, recombination path 0 1, nodes 2[0] 4[0], score: 5.8, displacement: 2\tATGCT\t1"
    g = api.GAFStruct.from_line(line)
    assert g.path == [1, 2, 4, 6, 7] and g.to_string() == line
    assert api.GAFStruct().to_string() == "\t0\t0\t0\t \t>0\t0\t0\t0\t0\t\t\t"     # GAFStruct::new() (gaf_output.rs:22-38)


def test_bad_inputs_are_status_codes_not_aborts(rg):
    from recgraph_amd import _lib, api
    with pytest.raises(_lib.RecGraphError):
        api.Graph.from_gfa_text("S\tx\tACGT\n")                       # non-numeric segment name
    with pytest.raises(_lib.RecGraphError):
        api.Graph.from_gfa_text("S\t1\tA\nS\t2\tC\nL\t1\t-\t2\t+\t0M\n")  # reverse orientation


def test_duplicate_segment_ids_are_rejected(rg):
    """`S 1 A` / `S 1 C` used to be accepted silently (flattened as two rows of one id): RG_ERR_GFA (-2)."""
    from recgraph_amd import _lib, api
    with pytest.raises(_lib.RecGraphError) as e:
        api.Graph.from_gfa_text("S\t1\tA\nS\t1\tC\n")
    assert e.value.code == -2 and "duplicate segment id 1" in str(e.value)
    with pytest.raises(_lib.RecGraphError):
        api.Graph.from_gfa_text("S\t1\tA\nS\t2\tC\nS\t3\tG\nS\t2\tT\nL\t1\t+\t2\t+\t0M\nL\t2\t+\t3\t+\t0M\n")


def test_option_switches(rg):
    """rg_set_option / rg_get_option: the diagnostic switches the GPU tests flip (no getenv on the run path)."""
    from recgraph_amd import _lib, api
    lib = _lib.load()
    for name in ("sweep_i32", "three_sweeps", "no_frec", "debug"):
        assert lib.rg_get_option(name.encode()) == 0
        api.set_option(name, 1)
        assert lib.rg_get_option(name.encode()) == 1
        api.set_option(name, 0)
    api.set_option("chunk_reads", 2048)
    assert lib.rg_get_option(b"chunk_reads") == 2048
    api.set_option("chunk_reads", 0)
    assert lib.rg_get_option(b"nope") == -1
    with pytest.raises(_lib.RecGraphError):
        api.set_option("nope", 1)


def test_graphs_that_are_not_topological_are_rejected(rg):
    """The DP kernels read rows above the current one only; a back-link would make them read rows never written
    (ADVICE r1).  RG_ERR_GRAPH (-4) on the host instead."""
    from recgraph_amd import _lib, api
    for gfa in ("S\t1\tA\nS\t2\tC\nL\t2\t+\t1\t+\t0M\n",                 # back-link
                "S\t1\tA\nS\t2\tC\nL\t1\t+\t2\t+\t0M\nL\t2\t+\t2\t+\t0M\n"):  # self-link
        with pytest.raises(_lib.RecGraphError) as e:
            api.Graph.from_gfa_text(gfa)
        assert e.value.code == -4
    ok = api.Graph.from_lnz("$AACAAAF", {1: [0], 3: [2], 4: [2], 5: [3, 4], 7: [6]})
    assert ok.rows == 8
    for preds in ({1: [0], 3: [4], 4: [2], 5: [3, 4], 7: [6]},     # later row
                  {1: [0], 3: [3], 7: [6]},                         # itself
                  {1: [0], 3: [7], 7: [6]},                         # the F row
                  {1: [0], 7: [7]},                                 # F from F
                  {1: [0], 3: [-1], 7: [6]}):
        with pytest.raises(_lib.RecGraphError) as e:
            api.Graph.from_lnz("$AACAAAF", preds)
        assert e.value.code == -4, preds


def test_gfa_without_a_usable_path_view_keeps_the_lnz_view(rg):
    """Modes 0-3 only need graph::read_graph (main.rs:29): a GFA whose P lines the pathwise kernels cannot take is
    still a graph; the reason comes back when a pathwise mode is requested."""
    from recgraph_amd import _lib, api
    base = "S\t1\tA\nS\t2\tC\nS\t3\tG\nL\t1\t+\t2\t+\t0M\nL\t1\t+\t3\t+\t0M\nL\t2\t+\t3\t+\t0M\n"
    cases = {
        "segment of row": base + "P\tp\t1+,3+\t*\n",                                   # segment 2 on no path
        "256 paths": base + "".join("P\tp%d\t1+,2+,3+\t*\n" % k for k in range(257)),
        "'+' path steps": base + "P\tp\t1+,2-,3+\t*\n",
        "unknown segment": base + "P\tp\t1+,9+\t*\n",
        "topological id order": base + "P\tp\t1+,3+,2+\t*\n",
    }
    for why, gfa in cases.items():
        g = api.Graph.from_gfa_text(gfa)
        assert g.rows == 5 and g.paths_number == 0 and why in g.path_error, (why, g.path_error)
        assert g.dump(0) == "$ACGF" and g.dump(2) == "1:0;2:1;3:1,2;4:3;"
        with pytest.raises(_lib.RecGraphError) as e:
            api.Batch(g, ["ACG"], api.make_params(api.MODE_PATHWISE))
        assert e.value.code == -4 and why in str(e.value)
    g = api.Graph.from_gfa_text(base + "P\tp\t1+,2+,3+\t*\nP\tq\t1+,3+\t*\n")
    assert g.paths_number == 2 and g.path_error == ""


def test_no_cpu_fallback(rg, example_gfa):
    """Without a HIP device the product fails loudly; it never routes through a CPU path."""
    from recgraph_amd import _lib, api
    if _lib.load().rg_device_count() > 0:
        pytest.skip("a GPU is present")
    g = api.Graph.from_gfa_text(example_gfa)
    with pytest.raises(_lib.RecGraphError) as e:
        api.align_batch(g, ["ACGT"], ["r"])
    assert e.value.code == -3
    # the streaming engine and the one-call multi-device entry too
    with pytest.raises(_lib.RecGraphError) as e:
        api.Stream(g, api.make_params(api.MODE_GLOBAL_POA))
    assert e.value.code == -3
    with pytest.raises(_lib.RecGraphError) as e:
        api.align_batch_multi(g, ["ACGT"], ["r"])
    assert e.value.code == -3


def test_product_never_imports_the_oracle():
    import subprocess
    import sys
    code = "import sys; import recgraph_amd; assert not any(m.startswith('oracle') for m in sys.modules), 'oracle imported'"
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "recgraph_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle/" not in txt and "liboracle" not in txt and "import oracle" not in txt, f


def test_wide_graphs_flatten_like_the_oracle(rg, oracle):
    """More than 64 paths (up to 256): path sets are four 64-bit words, groups are listed per 64-path page."""
    from recgraph_amd import api, synth
    for P, seed in ((65, 1), (100, 2), (130, 3), (256, 4)):
        t = synth.haplotype_graph(500, P, path_len=60, seed=seed).gfa()
        g, og = api.Graph.from_gfa_text(t), oracle.Graph.from_gfa_text(t)
        assert g.paths_number == P and g.path_error == ""
        for which in (10, 11, 12, 13, 14, 15, 16, 17, 18, 19):
            assert g.dump(which) == og.dump(which), (P, which)


def test_create_path_from_flat_arrays(rg):
    """rg_graph_create_path (what a caller holding a PathGraph passes: W = ceil(P / 64) mask words per row / edge) builds
    the same tables as the GFA route, for narrow and wide graphs."""
    from recgraph_amd import api, synth
    for P, seed in ((3, 1), (40, 2), (100, 3)):
        t = synth.haplotype_graph(300, P, path_len=50, seed=seed).gfa()
        g = api.Graph.from_gfa_text(t)
        lnz = g.dump(10)
        rows = g.dump(13).strip(";").split(";")
        row_paths = [{k for k, b in enumerate(bits) if b == "1"} for bits in rows]
        pred_hash = {}
        for item in g.dump(12).strip(";").split(";"):
            node, lst = item.split(":")
            pred_hash[int(node)] = {int(x.split("=")[0]): {k for k, b in enumerate(x.split("=")[1]) if b == "1"} for x in lst.split(",")}
        node_id = [int(x) for x in g.dump(15).split(",")]
        g2 = api.Graph.from_path_arrays(lnz, P, row_paths, pred_hash, node_id)
        assert g2.paths_number == P
        for which in (10, 11, 12, 13, 14, 15, 16, 17, 18, 19):
            assert g2.dump(which) == g.dump(which), (P, which)


def test_formatter_path_lists_replace_the_predhash_walk(rg, example_gfa):
    """The host formatter walks a path through per-path row lists instead of PredHash lookups (rg_gaf.cpp): the lists are
    checked step by step against those lookups when the graph is built; every graph family of the suite passes the check and
    lists exactly the rows of each path."""
    from recgraph_amd import api, synth
    texts = [example_gfa, synth.haplotype_graph(900, 7, path_len=150, seed=4).gfa(), synth.random_dag_graph(60, 6, seed=5).gfa()]
    texts += [v["gfa"] for v in VEC["path_graph"]]
    for t in texts:
        g = api.Graph.from_gfa_text(t)
        if not g.paths_number:
            continue
        d = g.dump(30).strip(";").split(";")
        assert d[0] == "ok"
        masks = g.dump(13).strip(";").split(";")
        for k, lst in enumerate(d[1:]):
            rows = [int(x) for x in lst.split(",")] if lst else []
            assert rows == [i for i in range(1, len(masks) - 1) if masks[i][k] == "1"]


def _parse_steps(text):
    head, recs, lead, _ = text.split(";", 3)[0], None, None, None
    parts = text.split(";")
    members = int(parts[0].split("=")[1])
    points = int(parts[1].split("=")[1])
    recs = [tuple(int(v, 16) for v in r.split(":")) for r in parts[2].split(",")] if parts[2] else []
    lead = {}
    if parts[3]:
        for e in parts[3].split(","):
            pt, k, m = e.split(":")
            lead[(int(pt), int(k))] = int(m, 16)
    return members, points, recs, lead


def _rec_fields(r):
    x, y, z, w = r
    return {"row": x & 0xfffff, "li": (x >> 20) & 7, "flags": (x >> 23) & 7, "field": (x >> 26) & 63, "slot": y & 0xfffff,
            "page": (y >> 29) & 3, "cont": bool(y >> 31), "mask": z | (w << 32)}


def test_step_tables_of_the_pathwise_sweeps():
    """rg_steps.cpp (host-only): the step tables k_sweep / k_sweep16 walk, their split form (TAIL groups behind their
    register runs) and the path-retirement tables, through rg_graph_dump 31-34.  Invariants the kernels rely on, on
    block-built and random-walk graphs with up to 64 paths and one with 70 (continuation entries, no retirement tables):
    every path of a row in exactly one group of the row; rows in sweep order; run fields; the split table holds the same
    records; a TAIL directly follows the last record of a run on the same paths; the retirement table equals its
    definition (the kernel's alpha rule) at every evaluation point and only shrinks."""
    from recgraph_amd import api, synth
    graphs = [synth.haplotype_graph(700, 8, path_len=120, seed=11), synth.haplotype_graph(2500, 32, path_len=260, seed=12),
              synth.random_dag_graph(150, 12, seed=13, max_jump=3, max_seg=8), synth.random_dag_graph(90, 40, seed=14, max_jump=6, max_seg=14),
              synth.random_dag_graph(60, 64, seed=15, similar=0.8), synth.random_dag_graph(80, 70, seed=16)]
    for sg in graphs:
        g = api.Graph.from_gfa_text(sg.gfa())
        P = g.paths_number
        row_masks = g.dump(13).strip(";").split(";")
        L = len(row_masks)
        paths_of = [sum(1 << k for k in range(P) if row_masks[i][k] == "1") for i in range(L)]
        total = sum(len(sg.path_sequence(k)) for k in range(P))
        for fwd, (cp, cs) in ((True, (31, 32)), (False, (33, 34))):
            members, points, plain, lead = _parse_steps(g.dump(cp))
            assert members == total == sum(bin(_rec_fields(r)["mask"]).count("1") for r in plain)
            f = [_rec_fields(r) for r in plain]
            rows = [r["row"] for r in f]
            assert rows == sorted(rows, reverse=not fwd) and set(rows) == set(range(1, L - 1))
            by_row = {}
            for r in f:
                by_row.setdefault(r["row"], []).append(r)
            for i, rs in by_row.items():
                seen = [0, 0, 0, 0]
                for r in rs:
                    assert r["mask"] and not (seen[r["page"]] & r["mask"])
                    seen[r["page"]] |= r["mask"]
                assert sum(m << (64 * pg) for pg, m in enumerate(seen)) == paths_of[i]
                plainflags = [r["flags"] for r in rs]
                if len(rs) > 1:
                    assert plainflags[0] & 1 and plainflags[-1] & 2 and all(not (fl & 4) for fl in plainflags)
                    assert all(not (fl & 1) for fl in plainflags[1:]) and all(not (fl & 2) for fl in plainflags[:-1])
                elif rs[0]["flags"] & 4:        # HEAD (4) or inner row (7) of a one-group run led by its lowest member
                    assert rs[0]["flags"] in (4, 7) and rs[0]["field"] >= 1 and not rs[0]["cont"]
            for t, r in enumerate(f):           # run field: rows left in the run, this one included, capped at 63
                if r["flags"] & 4:
                    nxt = f[t + 1] if t + 1 < len(f) else None
                    cont_run = nxt is not None and nxt["flags"] == 7
                    assert r["field"] == (min(nxt["field"] + 1, 63) if cont_run else 1), (t, r, nxt)
            if P > 64:
                # more than 64 paths: continuation entries, no split table; the retirement table (round 6) holds member sets of
                # several words — checked against its definition below with the groups reassembled from their entries
                assert any(r["cont"] for r in f)
                ev = 256
                assert points == len(f) // ev + 2
                cur, want, grp = {}, {}, 0
                for t in range(len(f) - 1, -1, -1):
                    r = f[t]
                    grp |= r["mask"] << (64 * r["page"])
                    if not r["cont"]:
                        if bin(grp).count("1") > 1:
                            a_bit = (r["mask"] & -r["mask"]).bit_length() - 1 if r["flags"] & 4 else r["field"]
                            alpha = 64 * r["page"] + a_bit
                            assert (grp >> alpha) & 1
                            cur[alpha] = cur.get(alpha, 0) | grp
                        grp = 0
                    if t % ev == 0:
                        for k, m in cur.items():
                            want[(t // ev, k)] = m
                assert lead == want and len(lead) > 10
                # the WIDE-RUN table (round 6; it takes the split table's place): the same records in the same order; the alpha
                # entry of a row with one group led by its lowest path is flagged HEAD / inner whatever continuation entries
                # follow, its run field counts the rows left in the segment over the continuation entries; the same lead table
                m2, p2, wide, lead2 = _parse_steps(g.dump(cs))
                fw = [_rec_fields(r) for r in wide]
                assert m2 == members and p2 == points and lead2 == want
                assert [(r["row"], r["slot"], r["mask"], r["page"], r["cont"]) for r in fw] == [(r["row"], r["slot"], r["mask"], r["page"], r["cont"]) for r in f]
                nflag = nspan = 0
                alphas = [t for t, r in enumerate(fw) if not r["cont"]]
                for a_i, t in enumerate(alphas):
                    r = fw[t]
                    rowrecs = [x for x in fw if x["row"] == r["row"]]
                    one_group = sum(1 for x in rowrecs if not x["cont"]) == 1
                    if r["flags"] & 4:
                        nflag += 1
                        nspan += len(rowrecs) > 1
                        assert one_group and r["flags"] in (4, 7) and r["field"] >= 1
                        assert all(x["page"] > r["page"] for x in rowrecs if x["cont"])
                        assert (r["mask"] & -r["mask"]).bit_length() - 1 == (f[t]["field"] if not (f[t]["flags"] & 4) else (r["mask"] & -r["mask"]).bit_length() - 1)
                        nxt = fw[alphas[a_i + 1]] if a_i + 1 < len(alphas) else None
                        cont_run = nxt is not None and nxt["flags"] == 7
                        assert r["field"] == (min(nxt["field"] + 1, 63) if cont_run else 1), (t, r, nxt)
                    else:
                        assert r["flags"] == f[t]["flags"] and r["field"] == f[t]["field"]
                assert nflag > 50 and nspan > 20, (nflag, nspan)
                continue
            # ---- split table ----
            m2, p2, split, lead2 = _parse_steps(g.dump(cs))
            fs = [_rec_fields(r) for r in split]
            assert m2 == members and sorted((r["row"], r["slot"], r["mask"]) for r in fs) == sorted((r["row"], r["slot"], r["mask"]) for r in f)
            ntails = 0
            for t, r in enumerate(fs):
                is_tail = bool(r["flags"] & 4) and r["field"] == 0
                if is_tail:
                    ntails += 1
                    prev = fs[t - 1]
                    assert prev["flags"] & 4 and prev["field"] == 1 and prev["mask"] == r["mask"] and prev["row"] != r["row"]
                    assert bin(r["mask"]).count("1") <= 4 and len(by_row[r["row"]]) > 1
            srows = {}
            for r in fs:
                srows.setdefault(r["row"], []).append(r)
            for i, rs in srows.items():
                if len(rs) > 1:              # first / last bits in the NEW order, exactly once each
                    assert sum(1 for r in rs if r["flags"] & 1) == 1 and rs[0]["flags"] & 1
                    assert sum(1 for r in rs if r["flags"] & 2) == 1 and rs[-1]["flags"] & 2
            # between the first and the last record of a row with several groups lie only REGISTER runs (<= 4 paths) and their
            # tails: the row's keys wait in the LDS words of the gather table meanwhile (k_sweep16: keys_ld / keys_st)
            pos_of = {}
            for t, r in enumerate(fs):
                pos_of.setdefault(r["row"], []).append(t)
            for i, ps in pos_of.items():
                if len(ps) > 1:
                    for t in range(ps[0], ps[-1] + 1):
                        r = fs[t]
                        if r["row"] != i:
                            assert r["flags"] & 4 and bin(r["mask"]).count("1") <= 4 and not r["cont"], (i, t, r)
            if sg is graphs[1]:
                assert ntails > 20           # the block-built graphs are what the split tables were made for
            # ---- retirement tables: the definition, with the kernel's alpha rule ----
            for recs, table, npts in ((f, lead, points), (fs, lead2, p2)):
                ev = 256
                assert npts == len(recs) // ev + 2
                cur = [0] * 64
                want = {}
                for t in range(len(recs) - 1, -1, -1):
                    r = recs[t]
                    if not r["cont"] and bin(r["mask"]).count("1") > 1:
                        alpha = (r["mask"] & -r["mask"]).bit_length() - 1 if r["flags"] & 4 else r["field"]
                        assert (r["mask"] >> alpha) & 1
                        cur[alpha] |= r["mask"]
                    if t % ev == 0:
                        for k in range(64):
                            if cur[k]:
                                want[(t // ev, k)] = cur[k]
                assert table == want
                for (pt, k), m in table.items():
                    assert pt == 0 or (table.get((pt - 1, k), 0) & m) == m


def test_f32_display_is_rusts_shortest_round_trip_text():
    """VERDICT r5 next #2c.  The recombination score is an f32 the reference prints with `{}` (recombination_output.rs:363-631,
    utils.rs:221-323): shortest decimal that round-trips, never scientific.  `rg::f32_display` (rg_gaf.cpp, std::to_chars) and
    the oracle's twin share one reading of that rule, so here the PRODUCT's formatter (compiled from rg_gaf.cpp into the host-only
    driver) meets an independent implementation — numpy's Dragon4 shortest-unique positional form — on every score the
    pipeline can print: (m + w) - (R + r * displacement) in f32 arithmetic, every k/10 of the score range, and random bit
    patterns."""
    import subprocess
    import numpy as np
    csrc = os.path.join(ROOT, "recgraph_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "asan"], stdout=subprocess.DEVNULL)
    vals = []
    tot = np.arange(-3000, 3001, dtype=np.float32)
    for R, r in ((4, 0.1), (10, 0.25), (0, 0.3)):
        for d in (0, 1, 2, 3, 7, 17, 100, 999, 2999):
            pen = np.float32(R) + np.float32(r) * np.float32(d)        # three separately rounded f32 operations, as the kernel does
            vals.append((tot - pen).astype(np.float32))
    vals.append((np.arange(-40000, 40001, dtype=np.float32) / np.float32(10)).astype(np.float32))
    rng = np.random.default_rng(6)
    rb = rng.integers(0, 1 << 32, size=20000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    vals.append(rb[np.isfinite(rb)])
    vals.append(np.array([0.0, -0.0, 1e-45, 3.4028235e38, 1.1754944e-38, 16777216.0, 0.1, 1e7, 1e-7], dtype=np.float32))
    v = np.concatenate(vals).astype(np.float32)
    bits = v.view(np.uint32)
    r = subprocess.run([os.path.join(csrc, "build", "host_asan"), "--f32"], input="".join("%08x\n" % b for b in bits),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = r.stdout.split("\n")[:-1]
    assert len(got) == len(v)
    for x, t in zip(v, got):
        exp = np.format_float_positional(x, unique=True, trim="-")
        assert t == exp, (x, t, exp)
        assert np.float32(t) == x or (x == 0 and float(t) == 0)          # round trip
        assert "e" not in t and "E" not in t
