# round 5, experiment 6: stream shapes at 3 waves per SIMD; per-lane thresholds (record growth)
mkdir -p gpurun_out/r05f
B="python bench.py --no-strong --no-cpu --no-probe"
W3=$PWD/tools/build/librecgraph_hip_FWDW3_REVW3.so
LM=$PWD/tools/build/librecgraph_hip_LANEMIN.so
run() { name=$1; shift; env "$@" > gpurun_out/r05f/$name.json 2>> gpurun_out/r05f/err_$name.log; }
run w2_h3_b4096 $B --steps 10 --warmup 3
run w3_h3_b4096 RG_LIB_PATH=$W3 $B --steps 10 --warmup 3
run w3_h4_b3072 RG_LIB_PATH=$W3 $B --steps 12 --warmup 4 --handles 4 --batch 3072
run w3_h5_b3072 RG_LIB_PATH=$W3 $B --steps 15 --warmup 5 --handles 5 --batch 3072
run w3_h2_b6144 RG_LIB_PATH=$W3 $B --steps 8 --warmup 2 --handles 2 --batch 6144
run w3_h6_b2048 RG_LIB_PATH=$W3 $B --steps 18 --warmup 6 --handles 6 --batch 2048
run w2_h4_b3072 $B --steps 12 --warmup 4 --handles 4 --batch 3072
run w3_h1_b3072 RG_LIB_PATH=$W3 $B --steps 4 --warmup 1 --handles 1 --batch 3072
run w3_h1_b6144 RG_LIB_PATH=$W3 $B --steps 4 --warmup 1 --handles 1 --batch 6144
run w2_h1_b4096 $B --steps 4 --warmup 1 --handles 1
run dbg_base RG_DEBUG=1 $B --steps 2 --warmup 0 --handles 1
run dbg_lanemin RG_DEBUG=1 RG_LIB_PATH=$LM $B --steps 2 --warmup 0 --handles 1
run lanemin_h3 RG_LIB_PATH=$LM $B --steps 10 --warmup 3
for f in gpurun_out/r05f/*.json; do python -c "
import json,sys; d=json.load(open('$f')); k=d['kernel_ms_per_step']; b=d['roofline']['reads_per_launch']; print('$f', round(d['value']), d['ms_per_step'], 'fwd', k.get('k_sweep16_fwd'), 'rev', k.get('k_sweep16_rev'), 'exp', k.get('k_expand'), 'cmr', k.get('k_colmax_rec'), k.get('k_colmax_rec_fwd'), 'reads/launch', b)"; done
grep "records:" gpurun_out/r05f/err_dbg_base.log | head -3; grep "records:" gpurun_out/r05f/err_dbg_lanemin.log | head -3
