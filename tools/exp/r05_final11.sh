# round 5, final tree (k_m0_simd changed last): GPU suite, config 2's profile pass and bench line
mkdir -p gpurun_out/r05x gpurun_out/r05v
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r05x/gpu_suite.txt
bash tools/profile_round.sh r05 "C2" > gpurun_out/r05_profile.log 2>&1
cp gpurun_out/r05/counters_C2.json profiles/
python bench.py --config C2 --steps 10 --warmup 3 > gpurun_out/r05v/r05_c2_bench.json 2> gpurun_out/r05v/C2.err
tail -n 2 gpurun_out/r05x/gpu_suite.txt
python -c "
import json; d=json.load(open('gpurun_out/r05v/r05_c2_bench.json')); r=d['roofline']; print(round(d['value']), d['ms_per_step'], d.get('parity_checked'), d.get('parity_ok'), r.get('frac'), r.get('counters_stale'), d['kernel_ms_per_step'])"
