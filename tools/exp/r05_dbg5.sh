python tools/exp/r05_dbg5.py C5 2>&1 | tail -6
python tools/exp/r05_dbg5.py C4 2>&1 | tail -6
RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_C_52840cd.so python tools/exp/r05_dbg5.py C5 2>&1 | tail -3
RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_C_52840cd.so python tools/exp/r05_dbg5.py C4 2>&1 | tail -3
