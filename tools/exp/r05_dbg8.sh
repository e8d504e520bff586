for v in FLATROWS FLATDIRS; do for i in 1 2; do RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_$v.so python tools/exp/r05_dbg5.py C4 2>&1 | grep checked; done; done
