"""Statistics build of k_sweep16 (RG_LIB_PATH = a library built with -DRG_SWEEP16_STALLSTAT[=2], tools/sweep_variants.sh
STALLSTAT / STALL2): shader-clock cycles the sweep waves of a config-5 batch spend waiting for row loads — at the start of
register runs and in the general path (=1), at the start of gather runs (=2) — against their whole life.
python tools/probes/stall_stat.py [reads] [config]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recgraph_amd import api, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
cfg = sys.argv[2] if len(sys.argv) > 2 else "C5"
sg, _, _ = synth.make_config(cfg, n_reads=1)
reads = synth.haplotype_reads(sg, n, 1000, seed=77, mosaic_frac=0.5 if cfg == "C5" else 0.0)
g = api.Graph.from_gfa_text(sg.gfa())
mode = api.MODE_RECOMBINATION if cfg == "C5" else api.MODE_PATHWISE
for rep in range(2):
    b = api.Batch(g, reads, api.make_params(mode))
    b.run(); b.fetch()
    c0, c1 = b.cell_updates, b.cell_updates_performed
    tot, gen = (c0 & 0xffffffff) << 8, (c0 >> 32) << 8
    run, nrun = (c1 & 0xffffffff) << 8, c1 >> 32
    print(cfg, os.path.basename(os.environ.get("RG_LIB_PATH", "default")), "reads", n,
          "| wave cycles per read (all sweeps) %.3g" % (tot / n),
          "| waits A: %.3g cycles per read = %.1f %%, %d events per read, %.0f cycles each" % (run / n, 100.0 * run / max(tot, 1), nrun / n, run / max(nrun, 1)),
          "| waits B (general path): %.3g per read = %.1f %%" % (gen / n, 100.0 * gen / max(tot, 1)), flush=True)
