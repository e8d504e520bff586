# where a sweep wave waits: run starts, general path, gather starts (statistics builds)
for v in STALLSTAT STALL2; do RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_$v.so python tools/probes/stall_stat.py 2048 C5 2>&1 | tail -2; done
RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_STALLSTAT.so python tools/probes/stall_stat.py 3072 C4 2>&1 | tail -2
