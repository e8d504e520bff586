# config 4: tile size against the 3 072 wave slots of the three-wave variant (six handles, final tree)
for b in 4096 3072 6144 4096 3072 6144; do
  python bench.py --config C4 --steps 12 --warmup 3 --no-cpu --no-strong --no-probe --batch $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $b', round(d['value']), d['ms_per_step'])"
done
