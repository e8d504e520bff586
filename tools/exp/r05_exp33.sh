# k_m0_simd: closed-form band (default) against the reference's loops (BANDLOOPS), same box, A/B/A/B
for v in BASE BANDLOOPS BASE BANDLOOPS BASE BANDLOOPS; do
  L=$PWD/tools/build/librecgraph_hip_$v.so; [ $v = BASE ] && L=$PWD/recgraph_amd/librecgraph_hip.so
  RG_LIB_PATH=$L python bench.py --config C2 --steps 10 --warmup 3 --no-cpu --no-strong 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v C2', round(d['value']), d['ms_per_step'], d['kernel_ms_per_step'])"
done
