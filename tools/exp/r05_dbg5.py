import sys, os
sys.path.insert(0, os.getcwd())
from recgraph_amd import api, synth
from oracle import oracle as O
cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
g, _, _ = synth.make_config(cfg, n_reads=1)
og = O.Graph.from_gfa_text(g.gfa())
gg = api.Graph.from_gfa_text(g.gfa())
mode, om = (api.MODE_RECOMBINATION, O.M8_ABS) if cfg == "C5" else (api.MODE_PATHWISE, O.M4_ABS)
reads = synth.haplotype_reads(g, 4096, 1000, seed=5683, mosaic_frac=0.5 if cfg == "C5" else 0.0)
names = ["read%d" % i for i in range(len(reads))]
check = list(range(0, 4096, 32))
_, _, exp = og.bench_text(om, [reads[i] for i in check], nthreads=8, name_prefix="x")
texts, _ = api.align_batch(gg, reads, names, mode=mode)
bad = []
for k, i in enumerate(check):
    e = exp[k].decode().replace("x%d\t" % k, "read%d\t" % i, 1)
    # (bench_text numbers the reads by their position in the subset: compare everything but the trailing read index)
    if texts[i].rsplit("\t", 1)[0] != e.rsplit("\t", 1)[0]:
        bad.append(i)
print(cfg, os.environ.get("RG_LIB_PATH", "current")[-28:], "checked", len(check), "mismatches", len(bad), bad[:12], flush=True)
for i in bad[:2]:
    k = check.index(i)
    a, b = texts[i].split("\t"), exp[k].decode().split("\t")
    for f in range(1, min(len(a), len(b)) - 1):
        if a[f] != b[f]:
            print("   read", i, "field", f, "gpu:", a[f][-120:], "| cpu:", b[f][-120:])
