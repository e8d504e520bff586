# round 5, experiment 2: LDS diet (13 KB per wave) — parity of the default build, then 2 vs 3 waves per SIMD
mkdir -p gpurun_out/r05b
timeout 1200 python -m pytest tests/test_gpu_api_surface.py tests/test_gpu_pathwise.py tests/test_gpu_full_size.py -x -q -m gpu > gpurun_out/r05b/pytest.log 2>&1
tail -5 gpurun_out/r05b/pytest.log
python bench.py --steps 4 --warmup 1 --no-strong --no-cpu --no-probe --handles 1 > gpurun_out/r05b/w2_h1.json 2>> gpurun_out/r05b/err.log
RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_FWDW3_REVW3.so python bench.py --steps 4 --warmup 1 --no-strong --no-cpu --no-probe --handles 1 > gpurun_out/r05b/w3_h1.json 2>> gpurun_out/r05b/err.log
python bench.py --steps 10 --warmup 3 --no-strong --no-cpu --no-probe > gpurun_out/r05b/w2_h3.json 2>> gpurun_out/r05b/err.log
RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_FWDW3_REVW3.so python bench.py --steps 10 --warmup 3 --no-strong --no-cpu --no-probe > gpurun_out/r05b/w3_h3.json 2>> gpurun_out/r05b/err.log
RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_FWDW3_REVW3.so python bench.py --config C4 --steps 10 --warmup 3 --no-strong --no-cpu --no-probe > gpurun_out/r05b/c4_w3_h3.json 2>> gpurun_out/r05b/err.log
python bench.py --config C4 --steps 10 --warmup 3 --no-strong --no-cpu --no-probe > gpurun_out/r05b/c4_w2_h3.json 2>> gpurun_out/r05b/err.log
for f in gpurun_out/r05b/*.json; do python -c "
import json,sys; d=json.load(open('$f')); k=d['kernel_ms_per_step']; print('$f', round(d['value']), d['ms_per_step'], k.get('k_sweep16_fwd'), k.get('k_sweep16_rev'), k.get('k_sweep16'))"; done
tail -3 gpurun_out/r05b/err.log
