"""Path retirement vs the speculation margin: member-row updates carried out / counted on config-5 reads.
   python tools/probes/margin_probe.py [reads]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recgraph_amd import api, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sg, reads, _ = synth.make_config("C5", n_reads=n)
g = api.Graph.from_gfa_text(sg.gfa())
for margin in (400, 160, 120, 80, 40, 0, -100):
    api.set_option("spec_margin", margin)
    b = api.Batch(g, reads, api.make_params(api.MODE_RECOMBINATION))
    b.run(); b.fetch()
    print("margin", margin, "performed / counted %.4f" % (b.cell_updates_performed / b.cell_updates), "counted", b.cell_updates)
api.set_option("spec_margin", 112)
