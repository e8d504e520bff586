#!/bin/bash
# Per-read instruction / cycle counters of one kernel family of a configuration (one rocprofv3 --pmc pass):
#   tools/probes/pmc_kernel.sh C2 k_m0_simd [reads per launch = 10000]
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
CFG=$1; KERN=$2; RPL=${3:-10000}
O=gpurun_out/pmc_probe_$CFG; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_ANY --kernel-trace --output-format csv -d $O -o p -- python3 bench.py --config $CFG --steps 2 --warmup 0 --no-cpu --no-strong --no-probe --handles 1 > /dev/null 2>&1
python3 - "$O" "$KERN" "$RPL" <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/p_counter_collection.csv", recursive=True)[0]
acc, n = collections.defaultdict(float), collections.Counter()
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(acc):
    print(k, "per read:", round(acc[k] / n[k] / float(sys.argv[3]), 1), "launches", n[k])
PY
