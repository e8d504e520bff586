# round 5, experiment 10: when a gather run pays (cost model constants)
mkdir -p gpurun_out/r05j
B="python bench.py --no-strong --no-cpu --no-probe --steps 12 --warmup 3"
run() { name=$1; shift; env "$@" > gpurun_out/r05j/$name.json 2>> gpurun_out/r05j/err.log; }
run base $B
for v in GR84x120 GR84x90 GR100x90; do run $v RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_$v.so $B; done
run base2 $B
run c4_base $B --config C4
for v in GR84x120 GR84x90 GR100x90; do run c4_$v RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_$v.so $B --config C4; done
for f in gpurun_out/r05j/*.json; do python -c "
import json,sys; d=json.load(open('$f')); k=d['kernel_ms_per_step']; print('$f', round(d['value']), d['ms_per_step'], 'fwd', k.get('k_sweep16_fwd'), 'rev', k.get('k_sweep16_rev'))"; done
