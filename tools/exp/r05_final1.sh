# round 5: whole GPU suite, then the profile pass of the final tree (kernel stats + counters of every configuration)
mkdir -p gpurun_out/r05p
timeout 1700 python -m pytest tests -x -q -m gpu > gpurun_out/r05p/pytest.log 2>&1
tail -3 gpurun_out/r05p/pytest.log
bash tools/profile_round.sh r05 "C5 C4 C3 C2" > gpurun_out/r05p/profile.log 2>&1
tail -3 gpurun_out/r05p/profile.log
ls gpurun_out/r05 | head -40
