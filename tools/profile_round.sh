#!/bin/bash
# Round profile pass on the GPU box (run from the repository root through gpurun):
#   tools/profile_round.sh TAG ["C5 C4 C3 C2"]
# calibrations (VALU issue rate, FETCH_SIZE / WRITE_SIZE units), then per config: rocprofv3 --kernel-trace --stats of a
# short bench and three separate --pmc passes; tools/make_counters.py digests everything into gpurun_out/TAG/.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
TAG=${1:-r02}
CFGS=${2:-"C5 C4 C3 C2"}
O=gpurun_out/$TAG
mkdir -p $O
if [ ! -x tools/build/valu_calib ]; then mkdir -p tools/build; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tools/build/valu_calib tools/valu_calib.hip || exit 1; fi
timeout 300 tools/build/valu_calib valu > $O/valu_calib_raw.json 2> $O/valu_calib.log
timeout 300 tools/build/valu_calib mem > $O/mem_calib_bytes.json 2>> $O/valu_calib.log
for grp in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/calib_pmc_$grp -o p -- tools/build/valu_calib mem > $O/calib_pmc_$grp.log 2>&1
done
for cfg in $CFGS; do
  c=$(echo $cfg | tr A-Z a-z)
  B=""; [ $cfg = C5 ] && B="--batch 2048"; [ $cfg = C4 ] && B="--batch 2048"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -o $TAG -- python3 bench.py --config $cfg --steps 4 --warmup 1 --no-cpu --no-strong --no-probe --handles 1 > $O/${TAG}_${c}_bench_under_rocprof.json 2>> $O/bench.err
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAVES"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_${c}_$i -o p -- python3 bench.py --config $cfg $B --steps 2 --warmup 0 --no-cpu --no-strong --no-probe --handles 1 > $O/pmc_${c}_bench.json 2> $O/pmc_${c}_$i.log
  done
done
python3 tools/make_counters.py $O $TAG
tail -3 $O/bench.err
