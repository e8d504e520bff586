"""GPU: alignment parity on the GFA shapes of tests/test_ingestion_cpu.py (several sinks, odd ids, shuffled lines)."""
import random

import pytest

from test_ingestion_cpu import odd_gfa, renumbered

pytestmark = pytest.mark.gpu


def test_alignment_parity_on_odd_gfas(oracle):
    from recgraph_amd import api, synth
    rnd = random.Random(7)
    cases = [(odd_gfa(), ["GACTT", "GTCA", "GTTT", "G", "GACCA", "TTTT", "GAC", "CA", "ACGTACGT"])]
    sg = synth.haplotype_graph(500, 6, path_len=100, seed=2)
    cases.append((renumbered(sg, rnd), synth.haplotype_reads(sg, 16, length=100, seed=3, mosaic_frac=0.5)))
    lg = synth.linear_graph(300, seed=5)
    cases.append((renumbered(lg, rnd), synth.substring_reads(lg, 16, 60, seed=4)))
    modes = ((api.MODE_GLOBAL_POA, "M0_SIMD"), (api.MODE_GLOBAL_POA_SCALAR, "M0_SCALAR"), (api.MODE_GAP_POA, "M2"),
             (api.MODE_LOCAL_POA, "M1_SIMD"), (api.MODE_GAP_LOCAL_POA, "M3"), (api.MODE_PATHWISE, "M4"), (api.MODE_PATHWISE_SEMI, "M5"),
             (api.MODE_RECOMBINATION, "M8_PRUNED"), (api.MODE_RECOMBINATION_SEMI, "M9_PRUNED"))
    for gfa, reads in cases:
        g, og = api.Graph.from_gfa_text(gfa), oracle.Graph.from_gfa_text(gfa)
        names = ["q%d" % i for i in range(len(reads))]
        for mode, om in modes:
            texts, status = api.align_batch(g, reads, names, mode=mode)
            for i, rd in enumerate(reads):
                exp, _, panic, _ = og.align(getattr(oracle, om), rd, name=names[i], idx=i + 1)
                if panic:
                    assert status[i] & api.READ_WOULD_PANIC, (om, i)
                else:
                    assert texts[i] == exp, (om, i, texts[i][-150:], exp[-150:])
