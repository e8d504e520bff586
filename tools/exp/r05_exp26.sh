# what path retirement is worth in the i32 sweep (one wave), and the striped long reads as they are
for o in "sweep_i32=1" "sweep_i32=1,no_retire=1"; do RG_REGION_OPTS=$o python tools/region_bench.py c5 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['case'], d.get('options'), d['reads_per_s'], d['ms_per_tile'], d['performed_over_counted'], d['parity_checked'], d['kernel_ms_per_tile'])"; done
python tools/region_bench.py len5000 2>&1 | tail -1 | cut -c1-900
