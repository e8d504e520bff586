#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/c2
for grp in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_BRANCH" "SQ_WAVES SQ_BUSY_CYCLES SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-50)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/c2/$tag -o p -- python3 bench.py --config C2 --steps 1 --warmup 0 --no-cpu > gpurun_out/c2/$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(float); n=collections.Counter()
for f in glob.glob('gpurun_out/c2/**/*counter_collection.csv', recursive=True):
    seen=set()
    for r in csv.DictReader(open(f)):
        if 'k_m0' in r['Kernel_Name']:
            acc[r['Counter_Name']] += float(r['Counter_Value'])
            seen.add(r['Dispatch_Id'])
    for c in set(r2 for r2 in acc): pass
    print(f.split('/')[2][:40], 'dispatches', len(seen))
wc = acc.get('SQ_WAVE_CYCLES', 1)
for k in sorted(acc): print('%-24s %.4g (%.1f%%)' % (k, acc[k], 100*acc[k]/wc))
PY
