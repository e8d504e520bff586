# round 5, experiment 12: tile sizes against the resident-wave counts (C4 at 3 waves per SIMD; the POA kernels), region cases
mkdir -p gpurun_out/r05l
B="python bench.py --no-strong --no-cpu --no-probe"
run() { name=$1; shift; env "$@" > gpurun_out/r05l/$name.json 2>> gpurun_out/r05l/err.log; }
for b in 4096 3072 6144; do for i in 1 2; do run c4_b${b}_$i $B --config C4 --steps 12 --warmup 3 --batch $b; done; done
for b in 10000 6144 12288; do run c2_b$b $B --config C2 --steps 12 --warmup 3 --batch $b; done
for b in 10000 5120 10240; do run c3_b$b $B --config C3 --steps 9 --warmup 3 --batch $b; done
for f in gpurun_out/r05l/c*.json; do python -c "
import json,sys; d=json.load(open('$f')); k=d['kernel_ms_per_step']; print('$f', round(d['value']), d['ms_per_step'], {a: round(b,2) for a,b in k.items()})"; done
python tools/region_bench.py len1500 x6 > gpurun_out/r05l/region.jsonl 2>> gpurun_out/r05l/err.log
cat gpurun_out/r05l/region.jsonl | cut -c1-400
