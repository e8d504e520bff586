# round 5, experiment 3: 2 vs 3 waves per SIMD after the LDS diet (flat loads of the profile fixed)
mkdir -p gpurun_out/r05c
B="python bench.py --no-strong --no-cpu --no-probe"
$B --steps 4 --warmup 1 --handles 1 > gpurun_out/r05c/w2_h1.json 2>> gpurun_out/r05c/err.log
RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_FWDW3_REVW3.so $B --steps 4 --warmup 1 --handles 1 > gpurun_out/r05c/w3_h1.json 2>> gpurun_out/r05c/err.log
RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_FWDW3_REVW3.so $B --steps 4 --warmup 1 --handles 1 --batch 3072 > gpurun_out/r05c/w3_h1_b3072.json 2>> gpurun_out/r05c/err.log
RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_FWDW3_REVW3.so $B --steps 4 --warmup 1 --handles 1 --batch 6144 > gpurun_out/r05c/w3_h1_b6144.json 2>> gpurun_out/r05c/err.log
$B --steps 10 --warmup 3 > gpurun_out/r05c/w2_h3.json 2>> gpurun_out/r05c/err.log
RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_FWDW3_REVW3.so $B --steps 10 --warmup 3 > gpurun_out/r05c/w3_h3.json 2>> gpurun_out/r05c/err.log
RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_REVW3.so $B --steps 10 --warmup 3 > gpurun_out/r05c/w2f_w3r_h3.json 2>> gpurun_out/r05c/err.log
RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_FWDW3_REVW3.so $B --steps 10 --warmup 3 --batch 6144 > gpurun_out/r05c/w3_h3_b6144.json 2>> gpurun_out/r05c/err.log
for f in gpurun_out/r05c/*.json; do python -c "
import json,sys; d=json.load(open('$f')); k=d['kernel_ms_per_step']; print('$f', round(d['value']), d['ms_per_step'], k.get('k_sweep16_fwd'), k.get('k_sweep16_rev'), k.get('k_sweep16'))"; done
tail -3 gpurun_out/r05c/err.log
