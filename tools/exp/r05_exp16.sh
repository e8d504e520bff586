# round 5, experiment 16: the i32 sweep's path retirement — whole GPU suite, HOXD70 / int32 leg
mkdir -p gpurun_out/r05q
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r05q/pytest.log 2>&1
tail -5 gpurun_out/r05q/pytest.log
python tools/region_bench.py hoxd70 > gpurun_out/r05q/region.jsonl 2>> gpurun_out/r05q/err.log
RG_NO_RETIRE=1 python tools/region_bench.py hoxd70 > gpurun_out/r05q/region_noretire.jsonl 2>> gpurun_out/r05q/err.log
cut -c1-330 gpurun_out/r05q/region.jsonl gpurun_out/r05q/region_noretire.jsonl
python bench.py --steps 10 --warmup 3 --no-cpu --no-strong > gpurun_out/r05q/c5_int32.json 2>> gpurun_out/r05q/err.log
python -c "
import json; d=json.load(open('gpurun_out/r05q/c5_int32.json')); print(d['value'], d['int32'])"
