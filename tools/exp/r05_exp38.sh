# k_m0_simd: the row's list flags from the host's table (default now): parity, then against the loops build as the same-box reference (it was 7.71-7.76 ms when the build before this one measured 7.14)
python -m pytest tests/test_gpu_m0.py tests/test_gpu_full_size.py tests/test_gpu_boundary.py tests/test_gpu_ingestion.py -x -q 2>&1 | tail -2
for v in BASE BANDLOOPS BASE BANDLOOPS BASE; do
  L=$PWD/tools/build/librecgraph_hip_$v.so; [ $v = BASE ] && L=$PWD/recgraph_amd/librecgraph_hip.so
  RG_LIB_PATH=$L python bench.py --config C2 --steps 10 --warmup 3 --no-cpu --no-strong 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v C2', round(d['value']), d['ms_per_step'], d['kernel_ms_per_step'])"
done
