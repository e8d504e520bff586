# k_m0_simd: rare-path addresses pinned inside their branch (default now) against the build with the reference's band loops (BANDLOOPS: 7.59-7.80 ms; closed form alone 7.40-7.49)
python -m pytest tests/test_gpu_m0.py tests/test_gpu_full_size.py tests/test_gpu_boundary.py -x -q -k "m0 or c2 or boundary" 2>&1 | tail -2
for v in BASE BANDLOOPS BASE BANDLOOPS BASE; do
  L=$PWD/tools/build/librecgraph_hip_$v.so; [ $v = BASE ] && L=$PWD/recgraph_amd/librecgraph_hip.so
  RG_LIB_PATH=$L python bench.py --config C2 --steps 10 --warmup 3 --no-cpu --no-strong 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v C2', round(d['value']), d['ms_per_step'], d['kernel_ms_per_step'])"
done
