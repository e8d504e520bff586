# config 5: handle count of the stream with the final kernels (fresh process each)
for h in 3 4 5 6 3 4 5 6; do
  python bench.py --steps 16 --warmup 4 --no-cpu --no-strong --no-probe --handles $h 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('handles $h', round(d['value']), d['ms_per_step'])"
done
for b in 3072 6144; do
  python bench.py --steps 16 --warmup 4 --no-cpu --no-strong --no-probe --batch $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $b', round(d['value']), d['ms_per_step'])"
done
