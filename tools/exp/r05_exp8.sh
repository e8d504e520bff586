# round 5, checkpoint: whole GPU suite + the four bench lines on the current tree
mkdir -p gpurun_out/r05h
timeout 1700 python -m pytest tests -x -q -m gpu > gpurun_out/r05h/pytest.log 2>&1
tail -3 gpurun_out/r05h/pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r05h/c5_bench.json 2> gpurun_out/r05h/c5_bench.err
for c in C4 C3 C2; do python bench.py --config $c --steps 10 --warmup 3 --no-cpu > gpurun_out/r05h/${c}_bench.json 2> gpurun_out/r05h/${c}_bench.err; done
for f in gpurun_out/r05h/*_bench.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', round(d['value']), d['ms_per_step'], d.get('int32'), d.get('strong_proxy'))"; done
