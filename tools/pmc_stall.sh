#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
B=${1:-1024}
mkdir -p gpurun_out/stall
for grp in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-60)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/stall/$tag -o p -- python3 bench.py --batch $B --steps 1 --warmup 0 --no-cpu > gpurun_out/stall/$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(float)
for f in glob.glob('gpurun_out/stall/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_sweep16' in r['Kernel_Name']: acc[r['Counter_Name']] += float(r['Counter_Value'])
wc = acc.get('SQ_WAVE_CYCLES', 1)
for k in sorted(acc): print('%-28s %.4g  (%.1f%% of wave cycles)' % (k, acc[k], 100 * acc[k] / wc))
PY
