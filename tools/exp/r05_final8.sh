# round 5, final tree: long randomised campaigns (the striped long reads of the campaign now retire paths too)
mkdir -p gpurun_out/r05x
python tools/fuzz_parity.py 900 801 > gpurun_out/r05x/r05_fuzz_801.txt 2>&1
RG_RETIRE_SHIFT=3 python tools/fuzz_parity.py 600 802 > gpurun_out/r05x/r05_fuzz_802_retire_every_8.txt 2>&1
tail -2 gpurun_out/r05x/r05_fuzz_80*.txt
