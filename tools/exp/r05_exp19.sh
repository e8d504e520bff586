# round 5, experiment 19: strong_proxy, repeats
mkdir -p gpurun_out/r05u
B="python bench.py --no-cpu --no-probe --steps 12 --warmup 4"
run() { name=$1; shift; env "$@" > gpurun_out/r05u/$name.json 2>> gpurun_out/r05u/err.log; }
for i in 1 2 3; do
run base_$i $B
run ramp1_$i $B --strong-ramp 1
run t2560_$i $B --strong-tile 2560
run t2560_ramp1_$i $B --strong-tile 2560 --strong-ramp 1
done
for f in gpurun_out/r05u/*.json; do python -c "
import json,sys; d=json.load(open('$f')); p=d['strong_proxy']; print('$f', round(d['value']), p['tiles'], p['ms'], p['reads_per_s'], p['ratio_vs_timed_region'])"; done
