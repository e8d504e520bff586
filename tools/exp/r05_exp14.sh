# round 5, experiment 14: k_opt0 on packed rows (cross-checked against the i32 form in RG_DEBUG mode), C4 with six handles
mkdir -p gpurun_out/r05n
timeout 1500 python -m pytest tests/test_gpu_pathwise.py tests/test_gpu_full_size.py -x -q -m gpu > gpurun_out/r05n/pytest.log 2>&1
tail -3 gpurun_out/r05n/pytest.log
B="python bench.py --no-strong --no-cpu --no-probe"
run() { name=$1; shift; env "$@" > gpurun_out/r05n/$name.json 2>> gpurun_out/r05n/err.log; }
run c5_1 $B --steps 12 --warmup 3
run c5_2 $B --steps 12 --warmup 3
run c5_h1 $B --steps 4 --warmup 1 --handles 1
run c4_1 $B --config C4 --steps 16 --warmup 4
run c4_2 $B --config C4 --steps 16 --warmup 4
for f in gpurun_out/r05n/c*.json; do python -c "
import json,sys; d=json.load(open('$f')); k=d['kernel_ms_per_step']; print('$f', round(d['value']), d['ms_per_step'], {a: round(b,2) for a,b in k.items() if b > 0.4})"; done
python tools/region_bench.py len600 m3x5 > gpurun_out/r05n/region.jsonl 2>> gpurun_out/r05n/err.log
cut -c1-300 gpurun_out/r05n/region.jsonl
