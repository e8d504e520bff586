# round 5: the bench lines of the final tree (driver's flags for C5; parity gate + CPU legs in every line)
mkdir -p gpurun_out/r05v
python bench.py --steps 20 --warmup 5 > gpurun_out/r05v/r05_c5_bench.json 2> gpurun_out/r05v/c5.err
for c in C4 C3 C2; do python bench.py --config $c --steps 10 --warmup 3 > gpurun_out/r05v/r05_$(echo $c | tr A-Z a-z)_bench.json 2> gpurun_out/r05v/$c.err; done
python tools/region_bench.py > gpurun_out/r05v/r05_region.jsonl 2> gpurun_out/r05v/region.err
for f in gpurun_out/r05v/r05_c*_bench.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; print('$f', round(d['value']), d['ms_per_step'], d.get('parity_checked'), d.get('parity_ok'), r.get('bound'), r.get('frac'), (r.get('valu') or {}).get('frac_step_clock'), (d.get('cpu_baseline') or {}).get('value'), (d.get('int32') or {}).get('reads_per_s'), (d.get('strong_proxy') or {}).get('ratio_vs_timed_region'))"; done
cut -c1-200 gpurun_out/r05v/r05_region.jsonl
