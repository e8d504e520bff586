# round 5, experiment 7: forward column maxima from the records (one record variant for both sweeps), kSemi template; 2 vs 3 waves
mkdir -p gpurun_out/r05g
timeout 1500 python -m pytest tests/test_gpu_pathwise.py tests/test_gpu_full_size.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r05g/pytest.log 2>&1
tail -3 gpurun_out/r05g/pytest.log
B="python bench.py --no-strong --no-cpu --no-probe"
W3=$PWD/tools/build/librecgraph_hip_REVW3.so
run() { name=$1; shift; env "$@" > gpurun_out/r05g/$name.json 2>> gpurun_out/r05g/err_$name.log; }
run w2_h3_b4096 $B --steps 10 --warmup 3
run w3_h3_b4096 RG_LIB_PATH=$W3 $B --steps 10 --warmup 3
run w3_h5_b3072 RG_LIB_PATH=$W3 $B --steps 15 --warmup 5 --handles 5 --batch 3072
run w3_h2_b6144 RG_LIB_PATH=$W3 $B --steps 8 --warmup 2 --handles 2 --batch 6144
run w2_h1_b4096 $B --steps 4 --warmup 1 --handles 1
run w3_h1_b6144 RG_LIB_PATH=$W3 $B --steps 4 --warmup 1 --handles 1 --batch 6144
run c4_w2_h3 $B --config C4 --steps 10 --warmup 3
run dbg RG_DEBUG=1 $B --steps 2 --warmup 0 --handles 1
for f in gpurun_out/r05g/*.json; do python -c "
import json,sys; d=json.load(open('$f')); k=d['kernel_ms_per_step']; b=d['roofline']['reads_per_launch']; print('$f', round(d['value']), d['ms_per_step'], 'fwd', k.get('k_sweep16_fwd'), 'rev', k.get('k_sweep16_rev'), k.get('k_sweep16'), 'exp', k.get('k_expand'), 'cmr', k.get('k_colmax_rec'), k.get('k_colmax_rec_fwd'), 'reads/launch', b)"; done
grep "records:" gpurun_out/r05g/err_dbg.log | head -3
