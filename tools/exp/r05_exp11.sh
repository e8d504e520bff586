# round 5, experiment 11: counted runs without peeks + src only with members; gather rule on C4 (lone kernels)
mkdir -p gpurun_out/r05k
timeout 1500 python -m pytest tests/test_gpu_pathwise.py tests/test_gpu_full_size.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r05k/pytest.log 2>&1
tail -3 gpurun_out/r05k/pytest.log
B="python bench.py --no-strong --no-cpu --no-probe"
G=$PWD/tools/build/librecgraph_hip_GR84x90.so
run() { name=$1; shift; env "$@" > gpurun_out/r05k/$name.json 2>> gpurun_out/r05k/err.log; }
run c5_h3 $B --steps 12 --warmup 3
run c5_h3_g RG_LIB_PATH=$G $B --steps 12 --warmup 3
run c5_h1 $B --steps 4 --warmup 1 --handles 1
run c5_h1_g RG_LIB_PATH=$G $B --steps 4 --warmup 1 --handles 1
for i in 1 2; do
run c4_h1_$i $B --config C4 --steps 6 --warmup 2 --handles 1
run c4_h1_g_$i RG_LIB_PATH=$G $B --config C4 --steps 6 --warmup 2 --handles 1
run c4_h3_$i $B --config C4 --steps 12 --warmup 3
run c4_h3_g_$i RG_LIB_PATH=$G $B --config C4 --steps 12 --warmup 3
done
python bench.py --config C2 --steps 10 --warmup 3 --no-cpu > gpurun_out/r05k/c2_full.json 2>> gpurun_out/r05k/err.log
for f in gpurun_out/r05k/c*.json; do python -c "
import json,sys; d=json.load(open('$f')); k=d['kernel_ms_per_step']; print('$f', round(d['value']), d['ms_per_step'], 'fwd', k.get('k_sweep16_fwd'), 'rev', k.get('k_sweep16_rev'), d.get('anchored'))"; done
tail -3 gpurun_out/r05k/err.log
