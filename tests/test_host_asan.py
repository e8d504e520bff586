"""CPU: the host-only text-eating code (GFA parser + flattening, FASTA parser, read canonicalisation, graph dumps) under
AddressSanitizer + UBSan (`make -C recgraph_amd/csrc asan`), on hand-made malformed inputs and seeded mutations of the
example graph / reads and of a small path graph.  Sanitizers run on the CPU build only (the GPU pool has no ASan)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_mutated_gfa_and_fasta_under_asan(tmp_path):
    csrc = os.path.join(ROOT, "recgraph_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "asan"], stdout=subprocess.DEVNULL)
    exe = os.path.join(csrc, "build", "host_asan")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    cases = [(os.path.join(HERE, "golden", "example_graph.gfa"), os.path.join(HERE, "golden", "example_reads.fa"), 600, 1)]
    from recgraph_amd import synth
    sg = synth.haplotype_graph(300, 5, path_len=80, seed=3)
    g2, f2 = tmp_path / "g.gfa", tmp_path / "r.fa"
    g2.write_text(sg.gfa())
    f2.write_text("".join(">r%d x\n%s\n" % (i, r) for i, r in enumerate(synth.haplotype_reads(sg, 9, 80, seed=4, mosaic_frac=0.5))))
    cases.append((str(g2), str(f2), 2500, 2))
    for gfa, fa, iters, seed in cases:
        r = subprocess.run([exe, gfa, fa, str(iters), str(seed)], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
        assert "gfa ok" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
        ok, rej = int(r.stdout.split()[2]), int(r.stdout.split()[4].rstrip(","))
        assert ok > 20 and rej > 20, r.stdout          # the mutations reach both outcomes
