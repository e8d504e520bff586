#!/bin/bash
cd "$GRAFT_REPO_ROOT/recgraph_amd/csrc" || exit 1
for k in 2 3 4 5; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DRG_SWEEP16_KRUN=$k -c rg_sweep16.hip -o build/rg_sweep16.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../librecgraph_hip.so $(ls build/*.o | grep -v stubs) -lpthread
  echo "== KRUN=$k"
  (cd "$GRAFT_REPO_ROOT" && timeout 120 python bench.py --steps 2 --warmup 1 --no-cpu 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k:v for k,v in d['kernel_ms_per_step'].items() if 'sweep' in k})")
done
