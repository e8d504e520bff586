# round 5, experiment 13: config 4 — handles per GPU (the -m 4 sweep runs three waves per SIMD: its small kernels wait for slots)
mkdir -p gpurun_out/r05m
B="python bench.py --no-strong --no-cpu --no-probe"
run() { name=$1; shift; env "$@" > gpurun_out/r05m/$name.json 2>> gpurun_out/r05m/err.log; }
for h in 3 4 5 6; do for i in 1 2 3; do run c4_h${h}_$i $B --config C4 --steps 16 --warmup 4 --handles $h; done; done
for h in 4 5; do for i in 1 2; do run c5_h${h}_$i $B --steps 12 --warmup 4 --handles $h --batch 3072; done; done
for f in gpurun_out/r05m/c*.json; do python -c "
import json,sys; d=json.load(open('$f')); k=d['kernel_ms_per_step']; print('$f', round(d['value']), d['ms_per_step'], {a: round(b,2) for a,b in k.items() if b > 0.5})"; done
python tools/region_bench.py len1500 > gpurun_out/r05m/region.jsonl 2>> gpurun_out/r05m/err.log
cut -c1-330 gpurun_out/r05m/region.jsonl
