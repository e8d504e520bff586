# round 5, final tree: profile pass of configs 5 and 4 (rg_pathwise.hip is part of their code hash), all bench lines, region, long reads
bash tools/profile_round.sh r05 "C5 C4" > gpurun_out/r05_profile.log 2>&1
cp gpurun_out/r05/counters_C5.json gpurun_out/r05/counters_C4.json profiles/
bash tools/exp/r05_final3.sh
mkdir -p gpurun_out/r05x
python tools/long_reads.py --modes 4,8 --check 4 > gpurun_out/r05x/r05_long_reads.jsonl 2> gpurun_out/r05x/long.err
cut -c1-200 gpurun_out/r05x/r05_long_reads.jsonl
