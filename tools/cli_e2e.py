#!/usr/bin/env python3
"""End-to-end timing of the CLI on BASELINE.json configs[4]: FASTA in, GAF out.

    python3 tools/cli_e2e.py [reads=102400] [out json]

Writes the synthetic graph (GFA) and the reads (FASTA, 25 seeded batches of 4096 like bench.py) to /tmp, runs
`python -m recgraph_amd.cli reads.fa graph.gfa -m 8 -R 4 -r 0.1 -B 1 > out.gaf` as a child process and prints ONE JSON
line: wall seconds of the whole process (interpreter start, FASTA parse, graph build, alignment of every read through the
streaming engine, GAF text to the file), reads/s, bytes written, the CLI's own "Done in" line."""
import json
import os
import resource
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from recgraph_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 102400
sg, _, _ = synth.make_config("C5", n_reads=1)
gfa, fa, out = "/tmp/c5_graph.gfa", "/tmp/c5_reads.fa", "/tmp/c5_out.gaf"
open(gfa, "w").write(sg.gfa())
t0 = time.time()
with open(fa, "w") as f:
    k = 0
    b = 0
    while k < n:
        reads = synth.haplotype_reads(sg, min(4096, n - k), 1000, seed=5678 + 5 + 100000 * (b + 1), mosaic_frac=0.5)
        for r in reads:
            f.write(">read%d\n%s\n" % (k, r))
            k += 1
        b += 1
gen_s = time.time() - t0
# VRAM freed by a process that just exited is scrubbed by the driver (~35 GB/s) and the next process's hipMalloc waits for
# it (profiles/r03_notes.md): measure the CLI on an idle device
time.sleep(float(os.environ.get("RG_E2E_IDLE", "15")))
t0 = time.time()
with open(out, "wb") as fo:
    p = subprocess.run([sys.executable, "-m", "recgraph_amd.cli", fa, gfa, "-m", "8", "-R", "4", "-r", "0.1", "-B", "1", "--timing"], stdout=fo,
                       stderr=subprocess.PIPE, cwd=ROOT)
wall = time.time() - t0
# peak resident set of the CLI process (the only child so far): the stream is bounded (--queue / --hold-mb) and the file is
# fed block by block, so it must not grow with the read set
rss_mb = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1024.0
lines = sum(1 for _ in open(out, "rb"))
print(json.dumps({"what": "python -m recgraph_amd.cli reads.fa graph.gfa -m 8 -R 4 -r 0.1 -B 1 > out.gaf (BASELINE configs[4])",
                  "reads": n, "wall_s": round(wall, 3), "reads_per_s": round(n / wall, 1), "gaf_lines": lines,
                  "gaf_bytes": os.path.getsize(out), "fasta_bytes": os.path.getsize(fa), "rc": p.returncode,
                  "cli_peak_rss_mb": round(rss_mb, 1),
                  "stderr": p.stderr.decode()[-1500:], "fasta_generation_s": round(gen_s, 1)}))
