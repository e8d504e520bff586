"""-m 0 parity on the GPU: HIP path (through the C ABI) vs the oracle, byte-for-byte stdout text."""
import pytest

pytestmark = pytest.mark.gpu


def _compare(oracle, og, graph, reads, names, mode_o, **kw):
    import recgraph_amd as rg
    from recgraph_amd import api
    texts, status = api.align_batch(graph, reads, names, mode=api.MODE_GLOBAL_POA, **kw)
    bad = []
    for i, rd in enumerate(reads):
        exp, score, panic, _ = og.align(mode_o, rd, name=names[i], idx=i + 1, **{k: v for k, v in kw.items() if k in ("b", "f", "bta")})
        if panic:
            assert status[i] & api.READ_WOULD_PANIC
            continue
        if texts[i] != exp:
            bad.append((i, texts[i][:300], exp[:300]))
    assert not bad, bad[:3]
    return texts


def test_m0_example_data(oracle, example_gfa, example_reads):
    from recgraph_amd import api
    names, reads = example_reads
    og = oracle.Graph.from_gfa_text(example_gfa)
    g = api.Graph.from_gfa_text(example_gfa)
    _compare(oracle, og, g, reads, names, oracle.M0_SIMD)


def test_m0_example_prefix_reads(oracle, example_gfa):
    """Reads that start at the source, so that real GAF lines (not 'band not enough') are produced."""
    import numpy as np
    from recgraph_amd import api, synth
    og = oracle.Graph.from_gfa_text(example_gfa)
    g = api.Graph.from_gfa_text(example_gfa)
    # walk the first paths of the example graph
    segs = {}
    paths = []
    for line in example_gfa.splitlines():
        f = line.split("\t")
        if f[0] == "S":
            segs[f[1]] = f[2]
        elif f[0] == "P":
            paths.append("".join(segs[s[:-1]] for s in f[2].split(",")))
    rng = np.random.default_rng(7)
    reads, names = [], []
    for k, p in enumerate(paths[:12]):
        for ln in (40, 150, 300, len(p)):
            s = list(p[:ln])
            for _ in range(max(1, ln // 60)):
                s[int(rng.integers(0, len(s)))] = "ACGT"[int(rng.integers(0, 4))]
            reads.append("".join(s))
            names.append(f"p{k}_{ln}")
    texts = _compare(oracle, og, g, reads, names, oracle.M0_SIMD)
    assert sum("band not enough" not in t for t in texts) >= len(texts) // 3
    # wider band (api.rs default bases_to_add = len * 0.1)
    _compare(oracle, og, g, reads[:16], names[:16], oracle.M0_SIMD, bta=15)


def test_m0_synthetic_c2(oracle):
    from recgraph_amd import api, synth
    sg, reads, _ = synth.make_config("C2", n_reads=256)
    # add reads anchored at the source so that full alignments exist too
    walk = sg.path_sequence(0)
    reads = reads + [walk[:150], walk[:151], walk[3:153], walk[:40], walk[:1]]
    names = ["r%d" % i for i in range(len(reads))]
    og = oracle.Graph.from_gfa_text(sg.gfa())
    g = api.Graph.from_gfa_text(sg.gfa())
    _compare(oracle, og, g, reads, names, oracle.M0_SIMD)
