# round 5, final tree (after the store-hazard fix): the profile pass, then the bench lines
bash tools/profile_round.sh r05 "C5 C4 C3 C2" > gpurun_out/r05_profile.log 2>&1
bash tools/exp/r05_final3.sh
