"""Deterministic synthetic inputs for the configurations of BASELINE.json.

Graphs: ids 1..S contiguous and topological, one source and one sink segment shared by every path,
every segment on at least one path, L lines sorted by (to, from), P lines in id order — so every
HashMap iteration order of the reference flattens them to the same arrays.

* ``linear_graph``  (configs 2, 3): a backbone with SNP / indel / multi-allelic bubbles, >= 85 % of the
  rows on the backbone.
* ``haplotype_graph`` (configs 4, 5): blocks of alternative alleles joined by shared segments; P
  haplotype walks of ~path_len bases, total rows ~= target (so ~rows/path_len parallel alleles).
"""
import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def _rand_seq(rng, n):
    return BASES[rng.integers(0, 4, size=n)].tobytes().decode()


def _mutate(rng, s):
    """One allele derived from another: a substitution, a short insertion or a short deletion."""
    s = list(s)
    r = rng.random()
    if r < 0.6 or len(s) < 2:
        k = int(rng.integers(0, len(s)))
        s[k] = "ACGT"[("ACGT".index(s[k]) + int(rng.integers(1, 4))) % 4]
    elif r < 0.8:
        k = int(rng.integers(0, len(s) + 1))
        s[k:k] = list(_rand_seq(rng, int(rng.integers(1, 4))))
    else:
        k = int(rng.integers(0, len(s) - 1))
        del s[k:k + int(rng.integers(1, min(3, len(s) - 1) + 1))]
    return "".join(s)


class SynthGraph:
    def __init__(self, segments, links, paths):
        self.segments = segments      # list of (id, seq), ids 1..S
        self.links = links            # list of (from, to)
        self.paths = paths            # list of lists of ids
        self.seq_of = {i: s for i, s in segments}

    @property
    def rows(self):
        return sum(len(s) for _, s in self.segments) + 2

    def gfa(self):
        out = ["H\tVN:Z:1.0"]
        for i, s in self.segments:
            out.append(f"S\t{i}\t{s}")
        for a, b in sorted(set(self.links), key=lambda l: (l[1], l[0])):
            out.append(f"L\t{a}\t+\t{b}\t+\t0M")
        for k, p in enumerate(self.paths):
            out.append(f"P\tpath{k}\t" + ",".join(f"{i}+" for i in p) + "\t*")
        return "\n".join(out) + "\n"

    def path_sequence(self, k):
        return "".join(self.seq_of[i] for i in self.paths[k])


def haplotype_graph(target_rows, n_paths, path_len=1000, seed=1234, shared_frac=0.3):
    """Blocks of alleles between shared segments (configs 4/5: 5 002 rows/16 paths, 10 002 rows/32 paths)."""
    rng = np.random.default_rng(seed)
    # mean alleles per block so that shared + alleles * variable ~= target rows
    variable = path_len * (1.0 - shared_frac)
    k_mean = max(2.0, (target_rows - path_len * shared_frac) / variable)
    segments, links, paths = [], [], [[] for _ in range(n_paths)]
    next_id = 1
    consumed = 0

    def add_seg(seq):
        nonlocal next_id
        segments.append((next_id, seq))
        next_id += 1
        return next_id - 1

    # source
    src = add_seg(_rand_seq(rng, int(rng.integers(4, 12))))
    for p in paths:
        p.append(src)
    consumed += len(segments[-1][1])
    prev_shared = src
    while consumed < path_len - 12:
        # allele block
        blen = int(rng.integers(1, 17))
        k = int(np.clip(round(rng.normal(k_mean, k_mean * 0.25)), 2, n_paths))
        base = _rand_seq(rng, blen)
        alleles = [base]
        for _ in range(k - 1):
            src_allele = alleles[int(rng.integers(0, len(alleles)))]
            a = _mutate(rng, src_allele)
            tries = 0
            while a in alleles and tries < 8:
                a = _mutate(rng, a)
                tries += 1
            if a in alleles:
                a = _rand_seq(rng, max(1, blen))
            alleles.append(a)
        ids = [add_seg(a) for a in alleles]
        # every allele on >= 1 path: first k paths (shuffled) take distinct alleles, the rest pick at random
        order = rng.permutation(n_paths)
        choice = {}
        for t, pid in enumerate(order):
            choice[int(pid)] = ids[t] if t < k else ids[int(rng.integers(0, k))]
        # shared segment after the block
        slen = int(rng.integers(1, max(2, int(2 * blen * shared_frac / (1 - shared_frac)) + 1)))
        sh = add_seg(_rand_seq(rng, slen))
        for pid in range(n_paths):
            a = choice[pid]
            links.append((prev_shared, a))
            links.append((a, sh))
            paths[pid] += [a, sh]
        prev_shared = sh
        consumed += blen + slen
    # sink
    snk = add_seg(_rand_seq(rng, max(4, path_len - consumed)))
    for p in paths:
        links.append((prev_shared, snk))
        p.append(snk)
    return SynthGraph(segments, links, paths)


def random_dag_graph(n_segments, n_paths, seed=1234, max_seg=10, max_jump=4, similar=0.5):
    """Segments in topological id order, every path a random walk source -> sink that skips up to ``max_jump - 1``
    segments per step: nested and overlapping bubbles, paths that share a segment and part ways after it, segments of
    one row, groups that are proper subsets of the row's paths — everything the block-shaped haplotype graphs never
    produce (step-table logic of the sweeps: heads, tails, gather runs, several groups per row).  ``similar``: chance that
    a segment is a mutated copy of an earlier one (alignments with real ties)."""
    rng = np.random.default_rng(seed)
    segments = []
    for i in range(1, n_segments + 1):
        if i > 2 and rng.random() < similar:
            base = segments[int(rng.integers(max(0, i - 6), i - 1))][1]
            seq = _mutate(rng, base) if len(base) > 1 else _rand_seq(rng, 1)
        else:
            seq = _rand_seq(rng, int(rng.integers(1, max_seg + 1)))
        segments.append((i, seq))
    paths = []
    for _ in range(n_paths):
        p, cur = [1], 1
        while cur < n_segments:
            cur = min(n_segments, cur + int(rng.integers(1, max_jump + 1)))
            p.append(cur)
        paths.append(p)
    used = {i for p in paths for i in p}
    for i in range(2, n_segments):
        if i not in used:           # every segment on at least one path: spliced into a random path at its place
            p = paths[int(rng.integers(0, n_paths))]
            at = next(t for t, x in enumerate(p) if x > i)
            p.insert(at, i)
    links = [(a, b) for p in paths for a, b in zip(p, p[1:])]
    return SynthGraph(segments, links, paths)


def linear_graph(target_rows, seed=1234):
    """Backbone with sparse bubbles (configs 2/3: ~1 002 / ~2 002 rows, >= 85 % of rows on the backbone)."""
    rng = np.random.default_rng(seed)
    segments, links = [], []
    next_id = 1
    rows = 0
    backbone_rows = 0

    def add_seg(seq):
        nonlocal next_id, rows
        segments.append((next_id, seq))
        rows += len(seq)
        next_id += 1
        return next_id - 1

    src = add_seg(_rand_seq(rng, int(rng.integers(6, 17))))
    backbone_rows += len(segments[-1][1])
    path_choices = [[src]]          # list of stages, each a list of alternative ids (first = backbone)
    prev = [src]
    while rows < target_rows - 30:
        r = rng.random()
        if r < 0.6:      # SNP
            ref = _rand_seq(rng, 1)
            alt = "ACGT"[("ACGT".index(ref) + int(rng.integers(1, 4))) % 4]
            alts = [ref, alt]
        elif r < 0.85:   # indel: alternative allele of 1-8 bases against a 1-base backbone allele
            alts = [_rand_seq(rng, 1), _rand_seq(rng, int(rng.integers(2, 9)))]
        else:            # 2-4 way multi-allelic
            alts = list({_rand_seq(rng, int(rng.integers(1, 4))) for _ in range(int(rng.integers(3, 5)))})
            if len(alts) < 2:
                alts.append(alts[0] + "A")
        ids = [add_seg(a) for a in alts]
        backbone_rows += len(alts[0])
        for p in prev:
            for a in ids:
                links.append((p, a))
        path_choices.append(ids)
        # backbone stretch long enough to keep >= 85 % of the rows on the backbone
        extra = sum(len(a) for a in alts[1:])
        stretch = int(rng.integers(max(8, 7 * extra), max(16, 9 * extra) + 8))
        cur = ids
        while stretch > 0:
            ln = int(min(stretch, rng.integers(1, 17)))
            s = add_seg(_rand_seq(rng, ln))
            backbone_rows += ln
            for a in cur:
                links.append((a, s))
            cur = [s]
            path_choices.append([s])
            stretch -= ln
        prev = cur
    snk = add_seg(_rand_seq(rng, max(4, target_rows - rows)))
    for p in prev:
        links.append((p, snk))
    path_choices.append([snk])
    n_paths = max(len(c) for c in path_choices)
    n_paths = max(n_paths, 2)
    paths = [[c[min(k, len(c) - 1)] if k < len(c) else c[0] for c in path_choices] for k in range(n_paths)]
    return SynthGraph(segments, links, paths)


def _apply_errors(rng, codes, sub=0.01, ins=0.001, dele=0.001):
    n = len(codes)
    r = rng.random(n)
    out = codes.copy()
    m = r < sub
    out[m] = (out[m] + rng.integers(1, 4, size=int(m.sum()))) % 4
    keep = ~((r >= sub) & (r < sub + dele))
    out = out[keep]
    k = int(rng.binomial(len(out), ins))
    if k:
        pos = np.sort(rng.integers(0, len(out) + 1, size=k))
        out = np.insert(out, pos, rng.integers(0, 4, size=k))
    return out


_CODE = np.zeros(256, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i


def haplotype_reads(graph, n_reads, length=1000, seed=5678, mosaic_frac=0.0):
    """Full walks trimmed/padded to exactly ``length`` bases; ``mosaic_frac`` of the reads switch
    haplotype once at a uniform position (config 5: 0.5)."""
    rng = np.random.default_rng(seed)
    pseq = [_CODE[np.frombuffer(graph.path_sequence(k).encode(), dtype=np.uint8)] for k in range(len(graph.paths))]
    # cumulative offsets of path steps, to switch haplotypes at a shared segment boundary
    reads = []
    for _ in range(n_reads):
        k = int(rng.integers(0, len(pseq)))
        s = pseq[k]
        if mosaic_frac > 0 and rng.random() < mosaic_frac:
            k2 = int(rng.integers(0, len(pseq)))
            frac = rng.random()
            a, b = pseq[k], pseq[k2]
            s = np.concatenate([a[:int(frac * len(a))], b[int(frac * len(b)):]])
        s = _apply_errors(rng, s)
        if len(s) >= length:
            s = s[:length]
        else:
            s = np.concatenate([s, rng.integers(0, 4, size=length - len(s)).astype(np.uint8)])
        reads.append(BASES[s].tobytes().decode())
    return reads


def substring_reads(graph, n_reads, length, seed=5678, anchored=False):
    """Substrings of a random source->sink walk at a uniform offset (configs 2/3); ``anchored``: at offset 0, i.e. reads
    that start at the graph's source (what a global alignment of a short read can actually place)."""
    rng = np.random.default_rng(seed)
    # stage alternatives recovered from the paths: walk = per stage a random alternative
    stages = list(zip(*graph.paths))
    stage_alts = [sorted(set(st)) for st in stages]
    reads = []
    for _ in range(n_reads):
        walk = "".join(graph.seq_of[a[int(rng.integers(0, len(a)))]] for a in stage_alts)
        codes = _CODE[np.frombuffer(walk.encode(), dtype=np.uint8)]
        if len(codes) > length:
            o = 0 if anchored else int(rng.integers(0, len(codes) - length + 1))
            codes = codes[o:o + length + 8]
        codes = _apply_errors(rng, codes)
        if len(codes) >= length:
            codes = codes[:length]
        else:
            codes = np.concatenate([codes, rng.integers(0, 4, size=length - len(codes)).astype(np.uint8)])
        reads.append(BASES[codes].tobytes().decode())
    return reads


def full_walk_reads(graph, n_reads, seed=5678):
    """Whole source->sink walks with the same error model (reads that a GLOBAL alignment can place inside a narrow band)."""
    rng = np.random.default_rng(seed)
    stage_alts = [sorted(set(st)) for st in zip(*graph.paths)]
    reads = []
    for _ in range(n_reads):
        walk = "".join(graph.seq_of[a[int(rng.integers(0, len(a)))]] for a in stage_alts)
        codes = _apply_errors(rng, _CODE[np.frombuffer(walk.encode(), dtype=np.uint8)])
        reads.append(BASES[codes].tobytes().decode())
    return reads


CONFIGS = {
    # name: (mode, graph builder, reads builder, n_reads of the full config)
    "C2": dict(mode=0, rows=1000, n=150, reads=10000),
    "C3": dict(mode=2, rows=2000, n=500, reads=10000),
    "C4": dict(mode=4, rows=5000, paths=16, n=1000, reads=50000),
    "C5": dict(mode=8, rows=10000, paths=32, n=1000, reads=100000),
}


def make_config(name, n_reads=None, graph_seed=1234):
    """Graph + reads of one BASELINE.json configuration (reads seed 5678 + config number)."""
    c = CONFIGS[name]
    num = int(name[1])
    nr = c["reads"] if n_reads is None else n_reads
    if name in ("C2", "C3"):
        g = linear_graph(c["rows"], seed=graph_seed)
        reads = substring_reads(g, nr, c["n"], seed=5678 + num)
    else:
        g = haplotype_graph(c["rows"], c["paths"], path_len=c["n"], seed=graph_seed)
        reads = haplotype_reads(g, nr, c["n"], seed=5678 + num, mosaic_frac=0.5 if name == "C5" else 0.0)
    return g, reads, c
