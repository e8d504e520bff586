# round 5, final tree (after the retirement across stripes): GPU suite, profile pass, bench lines, region, long reads
mkdir -p gpurun_out/r05x
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r05x/gpu_suite.txt
bash tools/profile_round.sh r05 "C5 C4 C3 C2" > gpurun_out/r05_profile.log 2>&1
cp gpurun_out/r05/counters_C*.json profiles/
bash tools/exp/r05_final3.sh
python tools/long_reads.py --modes 4,8 --check 4 > gpurun_out/r05x/r05_long_reads.jsonl 2> gpurun_out/r05x/long.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05x/smoke.txt 2>&1
tail -2 gpurun_out/r05x/gpu_suite.txt gpurun_out/r05x/smoke.txt; cut -c1-300 gpurun_out/r05x/r05_long_reads.jsonl
