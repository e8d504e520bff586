#!/bin/bash
cd "$GRAFT_REPO_ROOT/recgraph_amd/csrc" || exit 1
for d in "" "-DRG_EXP_NOTRACE"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $d -c rg_poa_banded.hip -o build/rg_poa_banded.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../librecgraph_hip.so $(ls build/*.o | grep -v stubs) -lpthread
  echo "== [$d]"
  (cd "$GRAFT_REPO_ROOT" && timeout 120 python bench.py --config C3 --steps 2 --warmup 1 --no-cpu 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms_per_step'])")
done
