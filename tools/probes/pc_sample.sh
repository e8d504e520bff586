#!/bin/bash
# PC sampling of one configuration's kernels (rocprofv3 --pc-sampling-beta-enabled): where the waves of k_sweep16 spend
# their time, instruction by instruction.  Run on the GPU box from the repository root:
#   tools/probes/pc_sample.sh [C5] [host_trap|stochastic] [interval] [unit]
# Writes gpurun_out/pcs_<cfg>_<method>/hist.json: samples per (kernel, code-object offset, instruction), digested on the box
# (the raw CSV is hundreds of MB).
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
CFG=${1:-C5}; METHOD=${2:-host_trap}; INTERVAL=${3:-1}; UNIT=${4:-time}
O=gpurun_out/pcs_${CFG}_${METHOD}; rm -rf $O; mkdir -p $O
RAW=$(mktemp -d /tmp/pcs_raw.XXXXXX)      # per run: an earlier run's CSVs must not be folded into this one's histogram
rocprofv3-avail list --pc-sampling > $O/avail.txt 2>&1 || rocprofv3-avail list > $O/avail.txt 2>&1
timeout 600 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $METHOD --pc-sampling-unit $UNIT --pc-sampling-interval $INTERVAL \
  --kernel-trace --output-format csv -d $RAW -o p -- python3 bench.py --config $CFG --steps 3 --warmup 1 --no-cpu --no-strong --no-probe --handles 1 \
  > $O/bench.json 2> $O/rocprof.log
echo "rocprofv3 exit $?" >> $O/rocprof.log
find $RAW -type f | head -50 > $O/files.txt
python3 - "$O" "$RAW" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
raw = sys.argv[2]
csv.field_size_limit(1 << 30)
kern = {}
for f in glob.glob(raw + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kern[r.get("Dispatch_Id")] = r.get("Kernel_Name", "")
res = {}
for f in glob.glob(raw + "/**/*pc_sampling*.csv", recursive=True):
    hist = collections.Counter()
    extra = collections.defaultdict(collections.Counter)
    cols = None
    n = 0
    for r in csv.DictReader(open(f)):
        if cols is None:
            cols = list(r.keys())
        n += 1
        k = kern.get(r.get("Dispatch_Id"), "?")
        key = (k[:60], r.get("Instruction", ""), r.get("Instruction_Comment", ""))
        hist[key] += 1
        for c in ("Stall_Reason", "Instruction_Type", "Wave_Issued", "Instruction_Not_Issued_Reason", "Stall_Reason_Not_Issued"):
            if c in r:
                extra[c][(k[:40], r[c])] += 1
    res[f.split("/")[-1]] = {
        "columns": cols, "samples": n,
        "hist": [[k[0], k[1], k[2], v] for k, v in hist.most_common(6000)],
        "extra": {c: [[a, b, v] for (a, b), v in d.most_common(200)] for c, d in extra.items()},
    }
json.dump(res, open(out + "/hist.json", "w"))
for name, d in res.items():
    print(name, d["samples"], d["columns"])
PY
tail -5 $O/rocprof.log
