# round 5, experiment 4: why is 3 waves per SIMD slower?  code (168 VGPRs) vs co-residency (cache footprint)
mkdir -p gpurun_out/r05d
export TMPDIR=/tmp
B="python bench.py --no-strong --no-cpu --no-probe --steps 4 --warmup 1 --handles 1"
W3=$PWD/tools/build/librecgraph_hip_FWDW3_REVW3.so
run() { name=$1; shift; env "$@" > gpurun_out/r05d/$name.json 2>> gpurun_out/r05d/err.log; }
run w2_b2048 $B --batch 2048
run w2_b4096 $B --batch 4096
run w3code_8waves_b2048 RG_LDS_PAD=6500 RG_LIB_PATH=$W3 $B --batch 2048
run w3code_10waves_b2560 RG_LDS_PAD=3000 RG_LIB_PATH=$W3 $B --batch 2560
run w3code_12waves_b3072 RG_LIB_PATH=$W3 $B --batch 3072
run w2code_6waves_b1536 RG_LDS_PAD=13000 $B --batch 1536
for f in gpurun_out/r05d/*.json; do python -c "
import json,sys; d=json.load(open('$f')); k=d['kernel_ms_per_step']; b=d['roofline']['reads_per_launch']; print('$f', round(d['value']), 'fwd', k.get('k_sweep16_fwd'), 'rev', k.get('k_sweep16_rev'), 'reads/ms fwd', round(b/k['k_sweep16_fwd'],1), 'rev', round(b/k['k_sweep16_rev'],1))"; done
# HBM-side traffic of the sweeps at 2 and 3 waves per SIMD
for v in w2 w3; do
  for grp in FETCH_SIZE WRITE_SIZE; do
    L=""; BT=2048; [ $v = w3 ] && L="$W3" && BT=3072
    RG_LIB_PATH=$L timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/r05d/pmc_${v}_$grp -o p -- python3 bench.py --no-strong --no-cpu --no-probe --steps 2 --warmup 0 --handles 1 --batch $BT > /dev/null 2>> gpurun_out/r05d/err.log
  done
done
python3 - <<'PY'
import csv, glob, collections
for v in ("w2", "w3"):
    for grp in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob("gpurun_out/r05d/pmc_%s_%s/**/p_counter_collection.csv" % (v, grp), recursive=True)
        acc, n = collections.defaultdict(float), collections.Counter()
        for f in fs:
            for r in csv.DictReader(open(f)):
                if "k_sweep16" in r["Kernel_Name"]:
                    key = r["Kernel_Name"][:40]
                    acc[key] += float(r["Counter_Value"]); n[key] += 1
        for k in acc:
            print(v, grp, k, "per launch", round(acc[k] / n[k]), "launches", n[k])
PY
rm -rf gpurun_out/r05d/pmc_*
tail -3 gpurun_out/r05d/err.log
