mkdir -p gpurun_out/r05s
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r05s/pytest.log 2>&1
tail -5 gpurun_out/r05s/pytest.log
