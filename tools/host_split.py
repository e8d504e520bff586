import sys, time; sys.path.insert(0, '.')
from recgraph_amd import api, synth
sg, _, _ = synth.make_config("C5", n_reads=1)
reads = synth.haplotype_reads(sg, 4096, 1000, seed=5683, mosaic_frac=0.5)
g = api.Graph.from_gfa_text(sg.gfa())
b = api.Batch(g, reads, api.make_params(8))
for it in range(3):
    t0 = time.perf_counter(); b.run(); t1 = time.perf_counter(); b.fetch(); t2 = time.perf_counter()
    txt = b.format_all(None, 1, 16); t3 = time.perf_counter()
    ks = sum(v[0] for v in b.kernel_stats().values())
    print("run %.1f ms (kernels %.1f) fetch %.1f format %.1f total %.1f" % ((t1-t0)*1e3, ks, (t2-t1)*1e3, (t3-t2)*1e3, (t3-t0)*1e3))
for nt in (8, 16, 32, 64, 128):
    t2 = time.perf_counter(); txt = b.format_all(None, 1, nt); t3 = time.perf_counter()
    print("threads", nt, "format %.1f ms" % ((t3-t2)*1e3), len(txt))
