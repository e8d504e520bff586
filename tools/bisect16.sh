#!/bin/bash
cd "$GRAFT_REPO_ROOT/recgraph_amd/csrc" || exit 1
for d in "-DRG16_OLD_BFI" "-DRG16_OLD_FOLD" "-DRG16_OLD_BFI -DRG16_OLD_FOLD"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $d -c rg_sweep16.hip -o build/rg_sweep16.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../librecgraph_hip.so build/*.o -lpthread
  echo "== $d"
  (cd "$GRAFT_REPO_ROOT" && timeout 100 python -m pytest tests/test_gpu_pathwise.py -x -q -k hand_derived 2>&1 | tail -2)
done
