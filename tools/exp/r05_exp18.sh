# round 5, experiment 18: the 1/8 share of the strong region on one GPU (strong_proxy): tile size and ramp
mkdir -p gpurun_out/r05t
B="python bench.py --no-cpu --no-probe --steps 20 --warmup 5"
run() { name=$1; shift; env "$@" > gpurun_out/r05t/$name.json 2>> gpurun_out/r05t/err.log; }
run base $B
run ramp1 $B --strong-ramp 1
run ramp2 $B --strong-ramp 2
run t2560 $B --strong-tile 2560
run t2048 $B --strong-tile 2048
run t2048_ramp1 $B --strong-tile 2048 --strong-ramp 1
run t4096 $B --strong-tile 4096
for f in gpurun_out/r05t/*.json; do python -c "
import json,sys; d=json.load(open('$f')); p=d['strong_proxy']; print('$f', round(d['value']), p['tiles'], p['ms'], p['reads_per_s'], p['ratio_vs_timed_region'], p['projected_speedup_at_8_gpus'])"; done
