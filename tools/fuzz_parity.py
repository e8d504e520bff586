"""Randomised parity campaign on the GPU: every mode vs the oracle on random graphs / reads / scores.
   python tools/fuzz_parity.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # noqa: E402
from recgraph_amd import api, synth      # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
HOX = open(os.path.join(ROOT, "tests", "golden", "HOXD55.mtx")).read()
PATH_MODES = [(api.MODE_PATHWISE, O.M4_ABS), (api.MODE_PATHWISE_SEMI, O.M5_ABS), (api.MODE_RECOMBINATION, O.M8_ABS),
              (api.MODE_RECOMBINATION_SEMI, O.M9_ABS)]
POA_MODES = [(api.MODE_GLOBAL_POA, O.M0_SIMD), (api.MODE_GLOBAL_POA_SCALAR, O.M0_SCALAR), (api.MODE_GAP_POA, O.M2),
             (api.MODE_LOCAL_POA, O.M1_SIMD), (api.MODE_LOCAL_POA_SCALAR, O.M1_SCALAR), (api.MODE_GAP_LOCAL_POA, O.M3)]
it = 0
checked = 0
fails = []
while time.time() < t_end and not fails:
    rng = np.random.default_rng(seed0 * 1000 + it)
    it += 1
    P = int(rng.choice([1, 2, 3, 5, 8, 16, 33, 40, 70, 130]))
    plen = int(rng.choice([20, 60, 150, 300, 700, 1300, 2600], p=[0.2, 0.2, 0.2, 0.2, 0.13, 0.05, 0.02]))
    if plen > 2047:
        P = min(P, 5)        # striped long reads: keep the oracle's L x n x P affordable
    rows = int(plen * rng.uniform(1.5, 6.0))
    if it % 2 == 1 and plen <= 1300:
        # every second configuration: random walks through segments in id order (nested / overlapping bubbles) instead of
        # allele blocks; the shortest path sets the read length
        sg = synth.random_dag_graph(max(8, int(plen / rng.uniform(1.5, 4.0))), P, seed=int(rng.integers(1, 10**6)),
                                    max_seg=int(rng.integers(2, 14)), max_jump=int(rng.integers(2, 7)), similar=float(rng.uniform(0, 0.9)))
        plen = min(len(sg.path_sequence(k)) for k in range(P))
    else:
        sg = synth.haplotype_graph(rows, P, path_len=plen, seed=int(rng.integers(1, 10**6)), shared_frac=float(rng.uniform(0.1, 0.6)))
    gfa = sg.gfa()
    nreads = int(rng.integers(3, 14))
    rl = int(rng.integers(max(2, plen // 3), plen + 20))
    reads = synth.haplotype_reads(sg, nreads, length=min(rl, plen), seed=int(rng.integers(1, 10**6)), mosaic_frac=float(rng.uniform(0, 1)))
    reads += ["ACGT"[int(x)] * int(rng.integers(1, 6)) for x in rng.integers(0, 4, size=2)]
    walk = sg.path_sequence(int(rng.integers(0, P)))
    reads.append(walk[:int(rng.integers(1, len(walk) + 1))])
    variant = int(rng.integers(0, 4))
    if variant == 0:
        sc, osc = None, None
    elif variant == 1:
        sc = api.create_score_matrix_i32(1, -1)
    elif variant == 2:
        sc = api.create_score_matrix_i32(3, -5)
    else:
        sc = {(k[0], k[1]): v for k, v in api.create_score_matrix_i32(matrix_file_path=os.path.join(ROOT, "tests", "golden", "HOXD55.mtx")).items()}
    osc = None if sc is None else O.scores_from_dict(sc)
    g = api.Graph.from_gfa_text(gfa)
    og = O.Graph.from_gfa_text(gfa)
    names = ["q%d" % i for i in range(len(reads))]
    modes = list(PATH_MODES) + [POA_MODES[int(rng.integers(0, 6))], POA_MODES[int(rng.integers(0, 6))]]
    for mode, om in modes:
        kw, okw = {}, {}
        if sc is not None:
            kw["score_matrix"] = sc; okw["scores"] = osc
        if mode in (api.MODE_RECOMBINATION, api.MODE_RECOMBINATION_SEMI):
            R, r, B = int(rng.choice([0, 2, 4, 9])), float(rng.choice([0.0, 0.1, 0.5])), float(rng.choice([1.0, 0.8, 0.5]))
            kw.update(R=R, r=r, B=B); okw.update(R=R, r=r, B=B)
        if mode in (api.MODE_GAP_POA, api.MODE_GAP_LOCAL_POA):
            o, e = int(rng.choice([0, -2, -4, -10])), int(rng.choice([-1, -2, -6]))
            kw.update(o=o, e=e); okw.update(o=o, e=e)
        if mode in (api.MODE_GLOBAL_POA, api.MODE_GLOBAL_POA_SCALAR, api.MODE_GAP_POA):
            b, f = float(rng.choice([1.0, 5.0, 30.0])), float(rng.choice([0.01, 0.1, 1.0]))
            kw.update(b=b, f=f); okw.update(b=b, f=f)
        rd = reads if mode in [m for m, _ in PATH_MODES] else [r_[:400] for r_ in reads]
        if os.environ.get("FUZZ_VERBOSE"):
            print("it", it, "P", P, "plen", plen, "rows", sg.rows, "mode", mode, "variant", variant, kw if sc is None else {k: v for k, v in kw.items() if k != "score_matrix"},
                  "reads", [len(q) for q in rd], file=sys.stderr, flush=True)
        texts, status = api.align_batch(g, rd, names, mode=mode, **kw)
        for i, q in enumerate(rd):
            exp, _, panic, _ = og.align(om, q, name=names[i], idx=i + 1, **okw)
            checked += 1
            if panic:
                if not status[i] & api.READ_WOULD_PANIC:
                    fails.append((it, mode, i, "expected panic", status[i]))
            elif texts[i] != exp:
                fails.append((it, mode, i, P, plen, len(q), variant, kw, texts[i][-160:], exp[-160:]))
        if fails:
            break
print("iterations", it, "alignments checked", checked, "failures", len(fails))
for f in fails[:3]:
    print(f)
sys.exit(1 if fails else 0)
