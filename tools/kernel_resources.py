#!/usr/bin/env python3
"""Register / scratch report of the HIP kernels (hipcc -Rpass-analysis=kernel-resource-usage, gfx950; no GPU needed).

    python3 tools/kernel_resources.py [file.hip ...] [-D...]      # default: rg_sweep16.hip; extra -D flags go to hipcc
    python3 tools/kernel_resources.py --isa [file.hip] [kernel name part ...] [-D...] [--json=out.json]
                                                                  # issue-class histogram of the kernels' loops (see isa_report)
    python3 tools/kernel_resources.py --hazards [file.hip ...] [-D...]
                                                                  # wide buffer stores whose data registers the NEXT instruction overwrites

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


_COMPILED = {}


def compile_once(src, defines=()):
    """(remark text, assembly lines) of one device-only hipcc run of `src` (cached per process: the register report and the hazard
    scan of a source share it — a compile of rg_sweep16.hip takes a minute)."""
    key = (src, tuple(defines))
    if key not in _COMPILED:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S",
                   "-Rpass-analysis=kernel-resource-usage", os.path.join(CSRC, src), "-o", out] + list(defines)
            r = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC, check=True)
            _COMPILED[key] = (r.stderr, open(out).read().split("\n"))
    return _COMPILED[key]


def report(src, defines=()):
    err = compile_once(src, defines)[0]
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


# ---- --isa: static issue-class histogram of a kernel's loops -------------------------------------------
# Classes follow profiles/valu_calib.json (measured on gfx950): `valu2` = the forms that issue in ~2.2 cycles per wave64
# instruction when two waves share a SIMD (plain VOP2 add / sub / and / or / xor / mov / ashr, v_bitop3_b32, compares,
# v_cndmask), `valu4` = everything else on the vector ALU (~4.1 cycles: every packed 16-bit, VOP3, shift-left, max / min,
# perm, DPP form), `lane` = v_readlane / v_readfirstlane / v_writelane (VALU issue slots; `spill` counts those that move an
# SGPR to or from a VGPR the compiler reserved for SGPR spills).
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_bitop3_b32", "v_mov_b32",
        "v_ashrrev_i32", "v_add_u16", "v_sub_u16", "v_max_i16", "v_max_u16", "v_min_i16", "v_min_u16", "v_cndmask_b32",
        "v_add_f32", "v_fma_f32", "v_add_co_u32", "v_sub_co_u32", "v_addc_co_u32", "v_subb_co_u32", "v_xnor_b32"}


def classify(mn, ops, spill_regs):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", mn)
    if mn.startswith("v_"):
        if base in ("v_readlane_b32", "v_readfirstlane_b32", "v_writelane_b32"):
            regs = set(re.findall(r"v\d+", ops))
            return "lane_spill" if regs & spill_regs else "lane"
        if mn.endswith("_dpp") or "row_shr" in ops or "row_bcast" in ops or "quad_perm" in ops or "wave_shr" in ops or "row_shl" in ops:
            return "valu4_dpp"
        if base.startswith("v_cmp"):
            return "valu2"
        if base in FAST:
            return "valu2"
        if base.startswith("v_pk_"):
            return "valu4_pk"
        return "valu4"
    if mn.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache")):
        return "smem"
    if mn.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
        return "wait"
    if mn.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_getpc")):
        return "branch"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if mn.startswith("scratch_"):
        return "scratch"
    if mn.startswith("ds_"):
        return "lds"
    return "other"


def isa_report(src, defines, want):
    """Per kernel whose demangled name contains one of `want`: totals and the loops (back edges) with their class histograms."""
    text = compile_once(src, defines)[1]
    starts = [(i, m.group(1)) for i, ln in enumerate(text) for m in [re.match(r"^(_Z[A-Za-z0-9_]+):", ln)] if m]
    names = demangle([n for _, n in starts])
    res = {}
    for (i, mangled) in starts:
        name = names.get(mangled, mangled).replace("(rg::SweepArgs)", "").replace("void ", "")
        if not any(w in name for w in want):
            continue
        end = next(j for j in range(i, len(text)) if text[j].startswith(".Lfunc_end"))
        body = text[i:end]
        spill_regs = set(re.findall(r"implicit-def: \$vgpr(\d+) : SGPR spill", "\n".join(body)))
        spill_regs = {"v" + r for r in spill_regs}
        insts, labels = [], {}
        for ln in body:
            m = re.match(r"^(\.LBB\d+_\d+):", ln)
            if m:
                labels[m.group(1)] = len(insts)
                continue
            m = re.match(r"^\t([a-z][a-z0-9_]+)\s*(.*?)(?:\s*;.*)?$", ln)
            if m and not m.group(1).startswith("."):
                insts.append((m.group(1), m.group(2)))
        cls = [classify(mn, ops, spill_regs) for mn, ops in insts]
        loops = []
        for k, (mn, ops) in enumerate(insts):
            if mn.startswith(("s_cbranch", "s_branch")):
                tgt = ops.strip().split()[-1] if ops.strip() else ""
                if tgt in labels and labels[tgt] <= k:
                    loops.append((labels[tgt], k, tgt))
        loops.sort()
        # innermost first reporting: a loop's own histogram covers its whole body (nested loops included)
        def hist(lo, hi):
            h = {}
            for c in cls[lo:hi + 1]:
                h[c] = h.get(c, 0) + 1
            return h
        rep = {"instructions": len(insts), "total": hist(0, len(insts) - 1), "sgpr_spill_vgprs": sorted(spill_regs), "loops": []}
        for lo, hi, tgt in loops:
            inner = [1 for a, b, _ in loops if (a, b) != (lo, hi) and lo <= a and b <= hi]
            rep["loops"].append({"label": tgt, "first": lo, "last": hi, "instructions": hi - lo + 1, "nested_loops": len(inner), "classes": hist(lo, hi)})
        res[name] = rep
    return res


def isa_main(argv):
    files = [a for a in argv if not a.startswith("-") and a.endswith(".hip")] or ["rg_sweep16.hip"]
    want = [a for a in argv if not a.startswith("-") and not a.endswith(".hip")] or ["k_sweep16<16, 2, true, false>", "k_sweep16<16, 0, true, false>", "k_sweep16<16, 0, false, false>"]
    defs = [a for a in argv if a.startswith("-D")]
    js = next((a.split("=", 1)[1] for a in argv if a.startswith("--json=")), None)
    allr = {}
    order = ["valu2", "valu4", "valu4_pk", "valu4_dpp", "lane", "lane_spill", "salu", "smem", "vmem", "lds", "scratch", "wait", "branch", "other"]
    for f in files:
        for name, rep in isa_report(f, defs, want).items():
            allr[name] = rep
            print("%s: %d instructions; SGPR-spill VGPRs %s" % (name, rep["instructions"], ",".join(rep["sgpr_spill_vgprs"]) or "none"))
            print("  %-28s %6s  %s" % ("region", "instr", "  ".join("%s" % c for c in order)))
            rows = [("whole kernel", rep["instructions"], rep["total"])] + [
                ("loop %s [%d..%d]%s" % (l["label"], l["first"], l["last"], " +%d nested" % l["nested_loops"] if l["nested_loops"] else ""), l["instructions"], l["classes"])
                for l in rep["loops"] if l["instructions"] >= 40]
            for label, n, h in rows:
                print("  %-28s %6d  %s" % (label[:28], n, "  ".join("%*d" % (len(c), h.get(c, 0)) for c in order)))
    if js:
        import json
        json.dump(allr, open(js, "w"), indent=1)


def _vregs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


WIDE_STORES = ("buffer_store_dwordx4", "buffer_store_dwordx3", "global_store_dwordx4", "global_store_dwordx3", "scratch_store_dwordx4",
               "scratch_store_dwordx3", "flat_store_dwordx4", "flat_store_dwordx3")
# (64-bit stores: the documented hazard starts above 64 bits and none was ever observed at 64; listed, not counted as hits)
STORES_64 = ("buffer_store_dwordx2", "global_store_dwordx2", "scratch_store_dwordx2", "flat_store_dwordx2")


def _valu_dests(u):
    """VGPRs a VALU instruction writes: its first operand, and for the VOP3 forms with a second destination (v_mad_u64_u32,
    v_div_scale ...: `vdst, sdst`) nothing more in VGPRs — carry-outs go to SGPR pairs / vcc.  v_swap_b32 writes both operands;
    DPP / SDWA forms write their first operand like everything else; v_readlane / v_readfirstlane / v_cmp write no VGPR."""
    mn = u.split()[0]
    if mn.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_nop")):
        return set()
    ops = [o.strip() for o in u[len(mn):].split(",")]
    d = _vregs(ops[0]) if ops else set()
    if mn.startswith("v_swap") and len(ops) > 1:
        d |= _vregs(ops[1])
    return d


def _data_operand(t):
    """data registers of a store: buffer_store: first operand; global / flat: second (after the address); scratch: second"""
    mn = t.split()[0]
    ops = [o.strip() for o in t[len(mn):].split(",")]
    if mn.startswith("buffer_store"):
        return _vregs(ops[0])
    return _vregs(ops[1]) if len(ops) > 1 else set()


def store_hazards(src, defines=(), want64=False):
    """gfx950: a VALU write to the data registers of a store of more than 64 bits within the ONE wait state behind it corrupts
    the store — also when the store takes its offset from an SGPR, the case LLVM's hazard recognizer exempts (found in round 5:
    rg_sweep16.hip, st_row: buffer stores; round 6 extends the scan to global / flat / SCRATCH stores — register spills are
    scratch_store_dwordx4 with an SGPR offset, and the register allocator reuses a spilled register at once — and to every
    kernel source).  Returns ([(kernel, line, store, overwriting instruction)], wide stores seen) over every kernel of `src`."""
    text = compile_once(src, defines)[1]
    hits, stores = scan_store_hazards(text, want64)
    names = demangle(sorted({h[0] for h in hits}))
    return [(names.get(k, k), ln, st, u) for k, ln, st, u in hits], stores


def scan_store_hazards(text, want64=False):
    """The scan itself, on the lines of an assembly listing: ([(mangled kernel, line, store, overwriting instruction)], stores seen)."""
    kinds = WIDE_STORES + (STORES_64 if want64 else ())
    hits, cur, stores = [], None, 0
    for i, ln in enumerate(text):
        m = re.match(r"^(_Z[A-Za-z0-9_]+):", ln)
        if m:
            cur = m.group(1)
        t = ln.strip()
        if not t.startswith(kinds):
            continue
        stores += 1
        data = _data_operand(t)
        j = i + 1
        while j < len(text):      # the next INSTRUCTION (labels, comments and directives are not wait states)
            u = text[j].split(";")[0].strip()
            j += 1
            if not u or u.startswith(".") or u.endswith(":"):
                continue
            if u.startswith("v_") and _valu_dests(u) & data:
                hits.append((cur, i + 1, t, u))
            break
    return hits, stores


def main():
    if "--hazards" in sys.argv:
        files = [a for a in sys.argv[1:] if not a.startswith("-")] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
        defs = [a for a in sys.argv[1:] if a.startswith("-D")]
        for f in files:
            hits, stores = store_hazards(f, defs, want64="--64" in sys.argv)
            print("%s: %d wide stores, %d with their data overwritten by the next instruction" % (f, stores, len(hits)))
            for h in hits:
                print("  %s  line %d: %s  ||  %s" % h)
        return
    if "--isa" in sys.argv:
        return isa_main([a for a in sys.argv[1:] if a != "--isa"])
    files = [a for a in sys.argv[1:] if not a.startswith("-")] or ["rg_sweep16.hip"]
    defs = [a for a in sys.argv[1:] if a.startswith("-")]
    for f in files:
        for k in report(f, defs):
            print("%-52s VGPRs %3d  spill V %3d S %3d  scratch %4d B/lane  occupancy %d  SGPRs %3d  LDS %d" % (
                k["name"][:52], k.get("VGPRs", -1), k.get("VGPRs Spill", -1), k.get("SGPRs Spill", -1),
                k.get("ScratchSize [bytes/lane]", -1), k.get("Occupancy [waves/SIMD]", -1), k.get("TotalSGPRs", -1), k.get("LDS Size [bytes/block]", -1)))


if __name__ == "__main__":
    main()
