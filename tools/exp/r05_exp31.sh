# config 5: the two runtime knobs of the pruning again with the final kernels (speculation margin, retirement period)
run() { RG_DEBUG=1 python bench.py --steps 16 --warmup 4 --no-cpu --no-strong --no-probe 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), d['ms_per_step'], 'performed/counted', round(d['cell_updates_performed_per_s']/d['cell_updates_per_s'],3), end=' ')"; grep "speculative bound" /tmp/err.txt | awk '{f+=$7; n+=$9} END {print "failed speculations", f, "of", n}'; }
for m in 112 96 80 64 48 112; do export RG_SPEC_MARGIN=$m; run "margin $m"; done
unset RG_SPEC_MARGIN
for k in 8 7 6 9 8; do export RG_RETIRE_SHIFT=$k; run "retire_shift $k"; done
