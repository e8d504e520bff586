# k_m0_simd: closed-form band alignment instead of the reference's three loops (scalar unit): parity, then config 2
python -m pytest tests/test_gpu_m0.py tests/test_gpu_full_size.py -x -q -k "m0 or c2" 2>&1 | tail -3
for i in 1 2 3; do python bench.py --config C2 --steps 10 --warmup 3 --no-cpu --no-strong 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2', round(d['value']), d['ms_per_step'], d['kernel_ms_per_step'])"; done
