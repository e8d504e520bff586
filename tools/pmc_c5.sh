#!/bin/bash
# PMC passes over the C5 bench (one counter group per pass, --kernel-trace only), summaries into gpurun_out/pmc/
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
B=${1:-2048}
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc/$tag -o p -- python3 bench.py --batch $B --steps 1 --warmup 0 --no-cpu > gpurun_out/pmc/$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob('gpurun_out/pmc/*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][:60]
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); 
        for k in acc:
            if 'sweep' in k: print(os.path.basename(os.path.dirname(d)), k, dict(acc[k]))
PY
