# the lone kernel durations outside the headline's shape (one handle)
RG_REGION_HANDLES=1 python tools/region_bench.py hoxd70 len1500 p128 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['case'], d['reads_per_s'], d['ms_per_tile'], d['kernel_ms_per_tile'])"
