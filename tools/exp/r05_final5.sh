# round 5: the C5 / C4 bench lines again, now that profiles/counters_C{4,5}.json are of this tree
mkdir -p gpurun_out/r05w
python bench.py --steps 20 --warmup 5 > gpurun_out/r05w/r05_c5_bench.json 2> gpurun_out/r05w/c5.err
python bench.py --config C4 --steps 10 --warmup 3 > gpurun_out/r05w/r05_c4_bench.json 2> gpurun_out/r05w/c4.err
python bench.py --config C4 --steps 10 --warmup 3 --no-cpu --no-strong > gpurun_out/r05w/r05_c4_bench_2.json 2>> gpurun_out/r05w/c4.err
for f in gpurun_out/r05w/r05_c*_bench*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; print('$f', round(d['value']), d['ms_per_step'], d.get('parity_checked'), d.get('parity_ok'), r.get('bound'), r.get('frac'), (r.get('valu') or {}).get('frac_step_clock'), (d.get('cpu_baseline') or {}).get('value'), (d.get('int32') or {}).get('reads_per_s'), (d.get('strong_proxy') or {}).get('ratio_vs_timed_region'))"; done
