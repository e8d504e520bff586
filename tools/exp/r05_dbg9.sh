for v in BUFWAIT LDGLC; do for i in 1 2; do RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_$v.so python tools/exp/r05_dbg5.py C4 2>&1 | grep checked; done; done
for v in FLATROWS BASE FLATROWS BASE; do
  L=$PWD/tools/build/librecgraph_hip_$v.so; [ $v = BASE ] && L=$PWD/recgraph_amd/librecgraph_hip.so
  RG_LIB_PATH=$L python bench.py --steps 12 --warmup 3 --no-cpu --no-strong --no-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v C5', d['value'], d['ms_per_step'], d.get('parity_ok'))"
done
for v in FLATROWS FLATROWS; do
  RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_$v.so python bench.py --config C4 --steps 12 --warmup 3 --no-cpu --no-strong --no-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v C4', d['value'], d['ms_per_step'], d.get('parity_ok'), d.get('parity_checked'))"
done
