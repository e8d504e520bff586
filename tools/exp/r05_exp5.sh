# round 5, experiment 5: buffer addressing, no block prefetch across row loops, row keys in LDS; 2 vs 3 waves
mkdir -p gpurun_out/r05e
timeout 1500 python -m pytest tests/test_gpu_pathwise.py tests/test_gpu_full_size.py -x -q -m gpu > gpurun_out/r05e/pytest.log 2>&1
tail -3 gpurun_out/r05e/pytest.log
B="python bench.py --no-strong --no-cpu --no-probe"
W3=$PWD/tools/build/librecgraph_hip_FWDW3_REVW3.so
RW3=$PWD/tools/build/librecgraph_hip_REVW3.so
run() { name=$1; shift; env "$@" > gpurun_out/r05e/$name.json 2>> gpurun_out/r05e/err.log; }
run w2_h1_b2048 $B --steps 4 --warmup 1 --handles 1 --batch 2048
run w3_h1_b3072 RG_LIB_PATH=$W3 $B --steps 4 --warmup 1 --handles 1 --batch 3072
run w3code_8w_b2048 RG_LDS_PAD=6500 RG_LIB_PATH=$W3 $B --steps 4 --warmup 1 --handles 1 --batch 2048
run w2_h3 $B --steps 10 --warmup 3
run w3_h3 RG_LIB_PATH=$W3 $B --steps 10 --warmup 3
run rw3_h3 RG_LIB_PATH=$RW3 $B --steps 10 --warmup 3
run w3_h3_b3072 RG_LIB_PATH=$W3 $B --steps 10 --warmup 3 --batch 3072
run w3_h2_b6144 RG_LIB_PATH=$W3 $B --steps 8 --warmup 2 --batch 6144 --handles 2
run c4_w2_h3 $B --config C4 --steps 10 --warmup 3
run c4_w3_h3 RG_LIB_PATH=$W3 $B --config C4 --steps 10 --warmup 3
for f in gpurun_out/r05e/*.json; do python -c "
import json,sys; d=json.load(open('$f')); k=d['kernel_ms_per_step']; b=d['roofline']['reads_per_launch']; print('$f', round(d['value']), d['ms_per_step'], 'fwd', k.get('k_sweep16_fwd'), 'rev', k.get('k_sweep16_rev'), k.get('k_sweep16'), 'reads/launch', b)"; done
tail -3 gpurun_out/r05e/err.log
