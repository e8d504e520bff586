# round 5: the profile pass of the final tree (kernel stats + counters of every configuration) and the bench lines
bash tools/profile_round.sh r05 "C5 C4 C3 C2" > gpurun_out/r05_profile.log 2>&1
tail -2 gpurun_out/r05_profile.log
