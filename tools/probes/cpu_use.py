import os, sys, time, resource
sys.path.insert(0, os.getcwd())
from recgraph_amd import api, synth
sg,_,_ = synth.make_config("C5", n_reads=1)
reads = synth.haplotype_reads(sg, 4096, 1000, seed=1, mosaic_frac=0.5)
g = api.Graph.from_gfa_text(sg.gfa())
packed = api.Batch.pack_reads(reads)
st = api.Stream(g, api.make_params(8), device_ids=[0], handles_per_device=3, tile_reads=4096, format_threads=4)
for _ in range(3): st.push(packed)
for _ in range(3): st.next()
r0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.time()
N = 30
for _ in range(N): st.push(packed)
for _ in range(N): st.next()
r1 = resource.getrusage(resource.RUSAGE_SELF); t1 = time.time()
print(os.environ.get("RG_LIB_PATH", "current"), "wall %.2f s, cpu user %.2f sys %.2f -> %.2f CPUs busy, %.0f reads/s" % (
    t1 - t0, r1.ru_utime - r0.ru_utime, r1.ru_stime - r0.ru_stime, (r1.ru_utime - r0.ru_utime + r1.ru_stime - r0.ru_stime) / (t1 - t0), N * 4096 / (t1 - t0)))
