mkdir -p gpurun_out/r05r
echo CURRENT; python tools/exp/r05_dbg1.py 2>&1 | tail -30
echo PREV; RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_PREV.so python tools/exp/r05_dbg1.py 2>&1 | tail -30
