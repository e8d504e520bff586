mkdir -p gpurun_out/r05a
python bench.py --steps 10 --warmup 3 --no-strong --no-cpu --no-probe > gpurun_out/r05a/base3h.json 2> gpurun_out/r05a/base3h.err
for pad in 0 7168 20480; do
RG_LDS_PAD=$pad python bench.py --steps 4 --warmup 1 --no-strong --no-cpu --no-probe --handles 1 > gpurun_out/r05a/occ_$pad.json 2>> gpurun_out/r05a/occ.err
done
bash tools/probes/pc_sample.sh C5 host_trap 1 time > gpurun_out/r05a/pcs_ht.log 2>&1
bash tools/probes/pc_sample.sh C5 stochastic 1048576 cycles > gpurun_out/r05a/pcs_st.log 2>&1
ls -la gpurun_out/r05a gpurun_out/pcs_*
timeout 900 python -m pytest tests/test_gpu_api_surface.py tests/test_gpu_pathwise.py -x -q -m gpu > gpurun_out/r05a/pytest.log 2>&1
tail -5 gpurun_out/r05a/pytest.log
