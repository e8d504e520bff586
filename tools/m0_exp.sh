#!/bin/bash
cd "$GRAFT_REPO_ROOT/recgraph_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DRG_EXP_COUNTFAST -c rg_poa.hip -o build/rg_poa.o 2>/dev/null
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../librecgraph_hip.so $(ls build/*.o | grep -v stubs) -lpthread
cd "$GRAFT_REPO_ROOT" && python3 - <<'PY'
import sys; sys.path.insert(0, ".")
from recgraph_amd import api, synth
sg, _, _ = synth.make_config("C2", n_reads=1)
reads = synth.substring_reads(sg, 100, 150, seed=5680)
g = api.Graph.from_gfa_text(sg.gfa())
b = api.Batch(g, reads, api.make_params(0)); b.run()
c = b.cell_updates
print("fast", c // 10**9 / 100, "not only_prev", c % 10**9 // 10**6 / 100, "not p_valid", c % 10**6 // 1000 / 100, "wide", c % 1000 / 100)
PY
