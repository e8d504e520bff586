"""Statistics build of k_sweep16 (RG_LIB_PATH = a library built with -DRG_SWEEP16_RETSTAT): per evaluation point of the path
retirement, how many paths are still needed and how many of them are not hopeless themselves (the others are kept because
they lead a needed path).  python tools/probes/retire_stat.py [reads]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recgraph_amd import api, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sg, reads, _ = synth.make_config("C5", n_reads=n)
g = api.Graph.from_gfa_text(sg.gfa())
for sweep, opt in (("forward", 2), ("reverse", 3)):
    api.set_option("no_retire", opt)
    b = api.Batch(g, reads, api.make_params(api.MODE_RECOMBINATION))
    b.run(); b.fetch()
    c0, c1 = b.cell_updates, b.cell_updates_performed
    api.set_option("no_retire", 0)
    ev, needed, alive = c0 & 0xffffffff, c0 >> 32, c1 & 0xffffffffffff
    print(sweep, "evaluations", ev, "needed paths per evaluation %.2f" % (needed / max(ev, 1)), "not hopeless %.2f" % (alive / max(ev, 1)))
