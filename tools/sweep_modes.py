import sys, time; sys.path.insert(0,'.')
from recgraph_amd import api, synth
sg, reads, c = synth.make_config("C5", n_reads=4096)
g = api.Graph.from_gfa_text(sg.gfa())
for mode in (4, 8):
    b = api.Batch(g, reads, api.make_params(mode)); b.run(); b.run()
    print(mode, {k: round(v[0],1) for k,v in b.kernel_stats().items()})
