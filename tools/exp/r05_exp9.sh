# round 5, experiment 9: outside the headline's shape (tools/region_bench.py), C = 32 variants
mkdir -p gpurun_out/r05i
timeout 900 python -m pytest tests/test_gpu_pathwise.py -x -q -m gpu -k "kilobase or longer or random_dag" > gpurun_out/r05i/pytest.log 2>&1
tail -3 gpurun_out/r05i/pytest.log
python tools/region_bench.py c5 x6 m3x5 hoxd70 len1500 len600 p128 m4_len1500 > gpurun_out/r05i/region.jsonl 2> gpurun_out/r05i/region.err
RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_G32.so python tools/region_bench.py len1500 > gpurun_out/r05i/region_g32.jsonl 2>> gpurun_out/r05i/region.err
RG_NO_RETIRE=1 python tools/region_bench.py len1500 > gpurun_out/r05i/region_noretire.jsonl 2>> gpurun_out/r05i/region.err
RG_SWEEP_I32=1 python tools/region_bench.py len1500 x6 > gpurun_out/r05i/region_i32.jsonl 2>> gpurun_out/r05i/region.err
for f in gpurun_out/r05i/*.jsonl; do echo $f; python -c "
import json,sys
for ln in open('$f'):
    d=json.loads(ln); k=d['kernel_ms_per_tile']; print(' ', d['case'], d['reads_per_s'], d['ms_per_tile'], d['sweep_kernels'], [k.get(x) for x in d['sweep_kernels']], d['parity_checked'])"; done
tail -3 gpurun_out/r05i/region.err
