#!/usr/bin/env python3
"""Digest the rocprofv3 outputs of tools/profile_round.sh into the small JSON/CSV files bench.py and DESIGN.md cite.

    python3 tools/make_counters.py <dir with the round's raw outputs> <TAG>

Writes into that directory (copy what should be judged into profiles/):
  valu_calib.json          issue-rate calibration (tools/valu_calib valu) + the peak bench.py prices VALU work against
  mem_calib.json           FETCH_SIZE / WRITE_SIZE units calibrated on known byte counts (tools/valu_calib mem)
  counters_<cfg>.json      per read per launch of the dominant kernel: calibrated fabric bytes, VALU wave-instructions
  <TAG>_<cfg>_pmc.csv      every counter of every kernel, summed over launches
  <TAG>_<cfg>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

O, TAG = sys.argv[1], sys.argv[2]
DOM = {"C2": "k_m0_simd", "C3": "k_poa_banded", "C4": "k_sweep", "C5": "k_sweep"}


def pmc_tables(pattern):
    """{kernel: {counter: (sum, launches)}} over every counter_collection.csv matching `pattern`."""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    nl = collections.defaultdict(lambda: collections.defaultdict(set))
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            nl[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    return {k: {c: (v, len(nl[k][c])) for c, v in cs.items()} for k, cs in acc.items()}


# ---- calibrations ----
vraw = os.path.join(O, "valu_calib_raw.json")
if os.path.exists(vraw):
    v = json.load(open(vraw))
    mix = v["sweep16 mix (pk_add, pk_max, pk_sub, pk_ashr, bitop3 x2, and_or, max_i32)"]
    peak = max(mix.values())
    cyc = v["compute_units"] * 4 * v["clock_mhz"] * 1e6 / peak
    json.dump({"peak_winstr_per_s": peak,
               "peak_definition": "best of 1/2/4 waves per SIMD of the sweep16 instruction mix stream (independent chains, no memory)",
               "cycles_per_wave64_instruction_per_simd": round(cyc, 3), "raw": v}, open(os.path.join(O, "valu_calib.json"), "w"), indent=1)
mem = {}
mj = os.path.join(O, "mem_calib_bytes.json")
if os.path.exists(mj):
    real = json.load(open(mj))
    t = pmc_tables(os.path.join(O, "calib_pmc_*", "**", "*counter_collection.csv"))
    for k, cs in t.items():
        if k in real:
            e = {}
            if "FETCH_SIZE" in cs and real[k]["read"]:
                e["fetch_bytes_per_counter_KB"] = real[k]["read"] / cs["FETCH_SIZE"][0]
            if "WRITE_SIZE" in cs and real[k]["written"]:
                e["write_bytes_per_counter_KB"] = real[k]["written"] / cs["WRITE_SIZE"][0]
            mem[k] = e
    json.dump({"note": "real bytes moved per unit of the rocprofv3 counter (the counters are documented in KB: 1024 = exact); "
                       "1 GiB buffers, past the 256 MiB Infinity Cache", "kernels": mem}, open(os.path.join(O, "mem_calib.json"), "w"), indent=1)

# correction factors for the sweep's access pattern (4 B/lane coalesced row loads/stores); fall back to the guide's x2 / x1
fcal = mem.get("calib_rmw_rows_4B", {}).get("fetch_bytes_per_counter_KB") or mem.get("calib_read_4B", {}).get("fetch_bytes_per_counter_KB") or 2048.0
wcal = mem.get("calib_rmw_rows_4B", {}).get("write_bytes_per_counter_KB") or mem.get("calib_write_4B", {}).get("write_bytes_per_counter_KB") or 1024.0

# ---- per config ----
for cfg in ("C2", "C3", "C4", "C5"):
    c = cfg.lower()
    t = pmc_tables(os.path.join(O, "pmc_%s_*" % c, "**", "*counter_collection.csv"))
    if not t:
        continue
    with open(os.path.join(O, "%s_%s_pmc.csv" % (TAG, c)), "w") as f:
        f.write("kernel,counter,sum_over_launches,launches\n")
        for k in sorted(t):
            for cn in sorted(t[k]):
                f.write('"%s",%s,%.1f,%d\n' % (k, cn, t[k][cn][0], t[k][cn][1]))
    bj = os.path.join(O, "pmc_%s_bench.json" % c)
    reads_per_launch = None
    if os.path.exists(bj):
        try:
            b = json.loads([ln for ln in open(bj) if ln.startswith("{")][-1])
            reads_per_launch = b["roofline"]["reads_per_launch"]
            kbase = b["roofline"]["kernel"]
            chash = b["roofline"].get("code_hash")
        except Exception:
            pass
    if not reads_per_launch:
        continue
    dom = {k: v for k, v in t.items() if DOM[cfg] in k}
    tot = collections.defaultdict(lambda: [0.0, 0])
    for k, cs in dom.items():
        for cn, (v, n) in cs.items():
            tot[cn][0] += v
            tot[cn][1] += n
    out = {"source": "profiles/%s_%s_pmc.csv (rocprofv3 --pmc, separate passes, %s reads per launch)" % (TAG, c, reads_per_launch),
           "kernel_base": kbase, "code_hash": chash, "kernels": sorted(dom), "reads_per_launch": reads_per_launch,
           "fetch_bytes_per_counter_KB": fcal, "write_bytes_per_counter_KB": wcal}
    if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
        fb = tot["FETCH_SIZE"][0] / tot["FETCH_SIZE"][1] * fcal
        wb = tot["WRITE_SIZE"][0] / tot["WRITE_SIZE"][1] * wcal
        out.update(fetch_bytes_per_launch=fb, write_bytes_per_launch=wb, hbm_bytes_per_read_per_launch=(fb + wb) / reads_per_launch)
    if "SQ_INSTS_VALU" in tot:
        out["valu_winstr_per_read_per_launch"] = tot["SQ_INSTS_VALU"][0] / tot["SQ_INSTS_VALU"][1] / reads_per_launch
    for cn in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY",
               "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_WAIT_INST_LDS",
               "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_INST_CYCLES_SALU", "SQ_WAVES"):
        if cn in tot:
            out[cn + "_per_launch"] = tot[cn][0] / tot[cn][1]
    json.dump(out, open(os.path.join(O, "counters_%s.json" % cfg), "w"), indent=1)
    for f in glob.glob(os.path.join(O, "prof_%s" % c, "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(O, "%s_%s_kernel_stats.csv" % (TAG, c)))
print(sorted(os.listdir(O)))
