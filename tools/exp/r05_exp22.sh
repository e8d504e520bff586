# register-run loop: the profile (LDS) wait at the top of every row (A) and the time inside the register-run blocks (B)
RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_STALL3.so python tools/probes/stall_stat.py 2048 C5 2>&1 | tail -2
