v=STNOP
for i in 1 2 3; do RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_$v.so python tools/exp/r05_dbg5.py C4 2>&1 | grep checked; done
RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_$v.so python tools/exp/r05_dbg5.py C5 2>&1 | grep checked
for c in C5 C4 C5 C4; do
  RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_$v.so python bench.py --config $c --steps 12 --warmup 3 --no-cpu --no-strong --no-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $c', d['value'], d['ms_per_step'], d.get('parity'))"
done
