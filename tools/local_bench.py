import sys, time; sys.path.insert(0, '.')
from recgraph_amd import api, synth
sg, _, _ = synth.make_config("C2", n_reads=1)
reads = synth.substring_reads(sg, 10000, 150, seed=5680)
g = api.Graph.from_gfa_text(sg.gfa())
for mode in (api.MODE_LOCAL_POA, api.MODE_LOCAL_POA_SCALAR, api.MODE_GAP_LOCAL_POA):
    b = api.Batch(g, reads, api.make_params(mode)); b.run()
    t0 = time.perf_counter(); b.run(); b.fetch(); x = b.format_all(None, 1, 16); t1 = time.perf_counter()
    print("mode", mode, "reads/s %.0f" % (len(reads) / (t1 - t0)), {k: round(v[0], 2) for k, v in b.kernel_stats().items()}, "cells", b.cell_updates)
