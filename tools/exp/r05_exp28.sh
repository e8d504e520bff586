# HOXD70 through the i32 pipeline: candidates per read, failed speculations
RG_DEBUG=1 RG_REGION_HANDLES=1 python tools/region_bench.py hoxd70 2>&1 | grep "^\[rg\]" | grep -v "buffers ready\|done after" | sort | uniq -c | sort -rn | head -12
RG_DEBUG=1 RG_REGION_HANDLES=1 RG_REGION_OPTS=sweep_i32=1 python tools/region_bench.py c5 2>&1 | grep "^\[rg\]" | grep -v "buffers ready\|done after" | sort | uniq -c | sort -rn | head -8
