#!/usr/bin/env python3
"""Register / scratch report of the HIP kernels (hipcc -Rpass-analysis=kernel-resource-usage, gfx950; no GPU needed).

    python3 tools/kernel_resources.py [file.hip ...] [-D...]      # default: rg_sweep16.hip; extra -D flags go to hipcc

Prints one line per kernel: VGPRs, spilled VGPRs / SGPRs, scratch bytes per lane, occupancy.  tests/test_kernel_resources.py
asserts that the headline variants of k_sweep16 use no scratch."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "recgraph_amd", "csrc")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def report(src, defines=()):
    with tempfile.TemporaryDirectory() as td:
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
               "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", os.path.join(td, "o.o")] + list(defines)
        err = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC).stderr
    kernels, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(.*?)\s+\[-Rpass-analysis", line)
        if not m:
            continue
        body = m.group(1)
        if body.startswith("Function Name:"):
            cur = {"mangled": body.split(":", 1)[1].strip()}
            kernels.append(cur)
        elif cur is not None and ":" in body:
            k, v = body.rsplit(":", 1)
            try:
                cur[k.strip()] = int(v.strip())
            except ValueError:
                cur[k.strip()] = v.strip()
    names = demangle([k["mangled"] for k in kernels])
    for k in kernels:
        k["name"] = names.get(k["mangled"], k["mangled"]).replace("(rg::SweepArgs)", "").replace("void ", "")
    return kernels


def main():
    files = [a for a in sys.argv[1:] if not a.startswith("-")] or ["rg_sweep16.hip"]
    defs = [a for a in sys.argv[1:] if a.startswith("-")]
    for f in files:
        for k in report(f, defs):
            print("%-52s VGPRs %3d  spill V %3d S %3d  scratch %4d B/lane  occupancy %d  SGPRs %3d  LDS %d" % (
                k["name"][:52], k.get("VGPRs", -1), k.get("VGPRs Spill", -1), k.get("SGPRs Spill", -1),
                k.get("ScratchSize [bytes/lane]", -1), k.get("Occupancy [waves/SIMD]", -1), k.get("TotalSGPRs", -1), k.get("LDS Size [bytes/block]", -1)))


if __name__ == "__main__":
    main()
