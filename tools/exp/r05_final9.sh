# round 5, final tree (after the closed-form band of k_m0_simd): GPU suite, config 2's profile pass and bench line
mkdir -p gpurun_out/r05x
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r05x/gpu_suite.txt
bash tools/profile_round.sh r05 "C2" > gpurun_out/r05_profile.log 2>&1
cp gpurun_out/r05/counters_C2.json profiles/
mkdir -p gpurun_out/r05v
python bench.py --config C2 --steps 10 --warmup 3 > gpurun_out/r05v/r05_c2_bench.json 2> gpurun_out/r05v/C2.err
python tools/fuzz_parity.py 240 901 > gpurun_out/r05x/r05_fuzz_901.txt 2>&1
tail -n 2 gpurun_out/r05x/gpu_suite.txt; tail -n 1 gpurun_out/r05x/r05_fuzz_901.txt
python -c "
import json; d=json.load(open('gpurun_out/r05v/r05_c2_bench.json')); r=d['roofline']; print(round(d['value']), d['ms_per_step'], d.get('parity_checked'), d.get('parity_ok'), r.get('frac'), r.get('counters_stale'), d['kernel_ms_per_step'])"
