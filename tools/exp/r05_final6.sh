# round 5, final tree: GPU suite, randomised campaigns, profile pass, bench lines
mkdir -p gpurun_out/r05x
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r05x/gpu_suite.txt
python tools/fuzz_parity.py 420 701 > gpurun_out/r05x/r05_fuzz_701.txt 2>&1
RG_RETIRE_SHIFT=4 python tools/fuzz_parity.py 240 702 > gpurun_out/r05x/r05_fuzz_702_retire_every_16.txt 2>&1
bash tools/profile_round.sh r05 "C5 C4 C3 C2" > gpurun_out/r05_profile.log 2>&1
cp gpurun_out/r05/counters_C*.json profiles/       # (the bench lines below read profiles/counters_*.json: code hash of THIS tree)
bash tools/exp/r05_final3.sh
tail -3 gpurun_out/r05x/*.txt
