# path retirement across stripes: parity first (with a hard time limit: a FIFO out of step would hang), then the long-read bench
timeout 900 python -m pytest tests/test_gpu_pathwise.py -x -q -k "longer_than_2047 or striped_long" 2>&1 | tail -5
timeout 600 python tools/long_reads.py --modes 8 --check 4 2>&1 | tail -3 | cut -c1-600
RG_NO_RETIRE=1 timeout 600 python tools/long_reads.py --modes 8 2>&1 | tail -1 | cut -c1-600
