# bench.py: the blocking-sync device flag of the N > 1 ranks, forced at N = 1 (does it work at all, does it cost anything)
for f in 0 1 0 1; do
  RG_BENCH_BLOCKING_SYNC=$f python bench.py --steps 16 --warmup 4 --no-cpu --no-strong --no-probe 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('blocking_sync $f', round(d['value']), d['ms_per_step'], 'host_cpus_busy', d['host_cpus_busy'])"; tail -2 /tmp/err.txt
done
