# retirement across stripes at full occupancy with a period of 8 records (every evaluation point makes the leading stripe wait): parity of 16 reads, no hang
RG_RETIRE_SHIFT=3 timeout 900 python tools/long_reads.py --modes 8 --reads 1024 --check 16 2>&1 | tail -1 | cut -c1-400
RG_RETIRE_SHIFT=5 RG_STRIPE_C=8 timeout 900 python tools/long_reads.py --modes 8 --reads 1024 --len 3500 --rows 7000 --paths 16 --check 16 2>&1 | tail -1 | cut -c1-400
