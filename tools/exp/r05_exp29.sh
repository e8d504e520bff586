# config 4: the distribution of the stream rate by handle count (the same library, fresh process each)
for h in 4 5 6 8; do for i in 1 2 3 4; do
  python bench.py --config C4 --steps 10 --warmup 3 --no-cpu --no-strong --no-probe --handles $h 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('handles $h', round(d['value']), d['ms_per_step'])"
done; done
