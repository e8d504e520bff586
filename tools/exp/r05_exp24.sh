# batched member fetches alone (NOAHEAD) against the build before them (STNOP), same box, lone-sweep probe times
for v in STNOP NOAHEAD STNOP NOAHEAD; do
  L=$PWD/tools/build/librecgraph_hip_$v.so
  RG_LIB_PATH=$L python bench.py --steps 12 --warmup 3 --no-cpu --no-strong 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$v C5', d['value'], d['ms_per_step'], k['k_sweep16_fwd'], k['k_sweep16_rev'])"
done
for v in STNOP NOAHEAD STNOP NOAHEAD; do
  L=$PWD/tools/build/librecgraph_hip_$v.so
  RG_LIB_PATH=$L python bench.py --config C4 --steps 12 --warmup 3 --no-cpu --no-strong 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$v C4', d['value'], d['ms_per_step'], k['k_sweep16_fwd'])"
done
