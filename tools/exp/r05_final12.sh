# round 5, final tree: three more randomised campaigns (default; i32 sweep forced; evaluation every 4 records)
mkdir -p gpurun_out/r05x
python tools/fuzz_parity.py 600 1001 > gpurun_out/r05x/r05_fuzz_1001.txt 2>&1
RG_SWEEP_I32=1 python tools/fuzz_parity.py 330 1002 > gpurun_out/r05x/r05_fuzz_1002_i32.txt 2>&1
RG_RETIRE_SHIFT=2 python tools/fuzz_parity.py 330 1003 > gpurun_out/r05x/r05_fuzz_1003_retire_every_4.txt 2>&1
tail -n 1 gpurun_out/r05x/r05_fuzz_1001.txt gpurun_out/r05x/r05_fuzz_1002_i32.txt gpurun_out/r05x/r05_fuzz_1003_retire_every_4.txt
