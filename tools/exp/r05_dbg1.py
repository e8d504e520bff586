import sys, os
sys.path.insert(0, os.getcwd())
from recgraph_amd import api, synth
g = synth.haplotype_graph(4200, 4, path_len=2600, seed=61)
rd = synth.haplotype_reads(g, 5, length=2600, seed=161, mosaic_frac=0.7)
rd += [g.path_sequence(3)[:2600 - 37], g.path_sequence(0)[:300], "ACGT" * 3]
gg = api.Graph.from_gfa_text(g.gfa())
names = ["r%d" % i for i in range(len(rd))]
base = {m: api.align_batch(gg, rd, names, mode=m)[0] for m in (api.MODE_PATHWISE, api.MODE_RECOMBINATION)}
for c in (8, 16, 32):
    api.set_option("stripe_c", c)
    for m in base:
        for rep in range(3):
            t = api.align_batch(gg, rd, names, mode=m)[0]
            bad = [i for i in range(len(rd)) if t[i] != base[m][i]]
            print("stripe_c", c, "mode", m, "rep", rep, "differs", bad, flush=True)
            for i in bad[:1]:
                a, b = t[i].split("\t"), base[m][i].split("\t")
                print("   fields differing:", [k for k in range(min(len(a), len(b))) if a[k] != b[k]], a[12][-60:] if len(a) > 12 else "", "|", b[12][-60:] if len(b) > 12 else "")
api.set_option("stripe_c", 0)
