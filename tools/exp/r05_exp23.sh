# profile a row ahead + batched member fetches (register runs): parity, then A/B against the build without (NOAHEAD has the batching too)
python -m pytest tests/test_gpu_pathwise.py tests/test_gpu_full_size.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
for v in BASE NOAHEAD BASE NOAHEAD; do
  L=$PWD/tools/build/librecgraph_hip_$v.so; [ $v = BASE ] && L=$PWD/recgraph_amd/librecgraph_hip.so
  RG_LIB_PATH=$L python bench.py --steps 12 --warmup 3 --no-cpu --no-strong 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$v C5', d['value'], d['ms_per_step'], k['k_sweep16_fwd'], k['k_sweep16_rev'])"
done
RG_LIB_PATH=$PWD/recgraph_amd/librecgraph_hip.so python bench.py --config C4 --steps 12 --warmup 3 --no-cpu --no-strong 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('C4', d['value'], d['ms_per_step'], k)"
