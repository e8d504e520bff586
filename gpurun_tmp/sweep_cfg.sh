for cfg in "3 0" "3 2048" "4 2048" "6 1024" "4 1024" "2 2048"; do
  set -- $cfg
  RG_CHUNK_READS=$2 python bench.py --steps 12 --warmup 2 --no-cpu --no-probe --handles $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('handles $1 chunk $2', d['value'], d['ms_per_step'], d['host_ms_per_step'])"
done
