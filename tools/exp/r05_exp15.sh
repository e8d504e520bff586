# round 5, experiment 15: speculative bound for the i32 one-wave sweep (HOXD70), abort test, int32 leg
mkdir -p gpurun_out/r05o
timeout 1500 python -m pytest tests/test_gpu_stream.py tests/test_gpu_pathwise.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r05o/pytest.log 2>&1
tail -3 gpurun_out/r05o/pytest.log
python tools/region_bench.py hoxd70 p128 > gpurun_out/r05o/region.jsonl 2>> gpurun_out/r05o/err.log
RG_NO_SPEC=1 python tools/region_bench.py hoxd70 > gpurun_out/r05o/region_nospec.jsonl 2>> gpurun_out/r05o/err.log
RG_DEBUG=1 python tools/region_bench.py hoxd70 2> gpurun_out/r05o/hoxd_debug.err > /dev/null
grep -c "did not reach" gpurun_out/r05o/hoxd_debug.err; grep "did not reach\|cand mean" gpurun_out/r05o/hoxd_debug.err | head -6
python bench.py --steps 10 --warmup 3 --no-cpu --no-strong > gpurun_out/r05o/c5_int32.json 2>> gpurun_out/r05o/err.log
python -c "
import json; d=json.load(open('gpurun_out/r05o/c5_int32.json')); print(d['value'], d['int32'])"
cut -c1-420 gpurun_out/r05o/region.jsonl gpurun_out/r05o/region_nospec.jsonl
