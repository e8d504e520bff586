#!/bin/bash
# Round profile pass on the GPU box: bench (with cpu baseline), rocprofv3 kernel stats, PMC traffic, other configs.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-r01c}
O=gpurun_out/$TAG
mkdir -p $O
timeout 600 python3 bench.py --steps 3 --warmup 1 > $O/${TAG}_c5_bench.json 2> $O/bench.err
for c in C2 C3 C4; do timeout 300 python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu > $O/${TAG}_$(echo $c | tr A-Z a-z)_bench.json 2>> $O/bench.err; done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $TAG -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $O/${TAG}_c5_bench_under_rocprof.json 2>> $O/bench.err
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- python3 bench.py --batch 2048 --steps 1 --warmup 0 --no-cpu > $O/pmc_$tag.log 2>&1
done
python3 - "$O" "$TAG" <<'PY'
import csv, glob, collections, os, sys, json
O, TAG = sys.argv[1], sys.argv[2]
rows = []
for d in sorted(glob.glob(O + '/pmc_*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen=set()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            key=(k, r['Dispatch_Id'])
            if key not in seen: seen.add(key); n[k]+=1
        for k in acc:
            for c, v in acc[k].items(): rows.append((k, c, v, n[k]))
with open(O + '/%s_c5_pmc_b2048.csv' % TAG, 'w') as f:
    f.write('kernel,counter,sum_over_launches,launches\n')
    for r in rows: f.write('"%s",%s,%.1f,%d\n' % r)
sw = {}
for k, c, v, n in rows:
    if 'k_sweep16' in k:
        pv, pn = sw.get(c, (0.0, 0))
        sw[c] = (pv + v, pn + n)
if 'FETCH_SIZE' in sw and 'WRITE_SIZE' in sw:
    fb = sw['FETCH_SIZE'][0] * 1024 / sw['FETCH_SIZE'][1]; wb = sw['WRITE_SIZE'][0] * 1024 / sw['WRITE_SIZE'][1]
    json.dump({"kernel": "rg::k_sweep16<16>", "reads_per_launch": 2048, "fetch_bytes_per_launch_raw": fb, "write_bytes_per_launch_raw": wb,
               "fetch_correction": "x2 (MI355X_MICROARCH.md HBM section: gfx950 FETCH_SIZE counts 128-B requests as 64 B; calibrated there for 16 B/lane streams, this kernel issues 4 B/lane coalesced dwords: treated the same, uncalibrated)",
               "hbm_bytes_per_read_per_launch": (2 * fb + wb) / 2048}, open(O + '/traffic_C5.json', 'w'), indent=1)
for f in glob.glob(O + '/prof/**/*kernel_stats.csv', recursive=True):
    os.system('cp %s %s/%s_c5_kernel_stats.csv' % (f, O, TAG))
PY
ls $O; tail -c 600 $O/${TAG}_c5_bench.json; cat $O/bench.err | tail -5
