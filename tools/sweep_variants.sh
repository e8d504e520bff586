#!/bin/bash
# Timing-only variants of k_sweep16 ("make the evidence reproducible"): builds one library per
# -DRG_SWEEP16_* flag into tools/build/ (the results of these builds are garbage: nothing ships or tests them) and times
# the config-5 sweeps of each with a one-handle stream (kernel durations = HIP events, nothing else on the GPU).
# Two VALID variants too: RSH<k> (path retirement evaluated every 2^k records instead of 256: RSH4 makes the small graphs of the
# tests and of tools/fuzz_parity.py retire paths) and RETSTAT (statistics build for tools/probes/retire_stat.py).
#   HERE (no GPU):   tools/sweep_variants.sh build
#   on the GPU box:  tools/sweep_variants.sh run > gpurun_out/sweep_variants.txt
cd "$(dirname "$0")/.." || exit 1
VARIANTS="${VARIANTS:-BASE NOKEYS NOEMIT NODIRS NOKEYS_NOEMIT NOROWS_NOEMIT NOEMIT_NOROWS32}"
if [ "$1" = build ]; then
  mkdir -p tools/build
  for v in $VARIANTS; do
    flags=""; [ $v != BASE ] && for f in ${v//_/ }; do
      case $f in PFD1) flags="$flags -DRG_SWEEP16_PFD=1";; PFDFWD) flags="$flags -DRG_SWEEP16_PFD_FWD=1";; KRUNNOST|KRUNNOLD) flags="$flags -DRG_SWEEP16_$f";; NORUNWAIT) flags="$flags -DRG_SWEEP16_RUNWAIT=0";; CHAIN0) flags="$flags -DRG_SWEEP16_CHAIN=0";; CHAIN2) flags="$flags -DRG_SWEEP16_CHAIN=2";; PF1) flags="$flags -DRG_SWEEP16_PF=1";; KRUN*) flags="$flags -DRG_SWEEP16_KRUN=${f#KRUN}";; THRLDS0) flags="$flags -DRG_SWEEP16_THRLDS=0";; REVK*) flags="$flags -DRG_SWEEP16_KRUN_REV=${f#REVK}";; REVW*) flags="$flags -DRG_SWEEP16_REV_WAVES=${f#REVW}";; FWDW*) flags="$flags -DRG_SWEEP16_FWD_WAVES=${f#FWDW}";; LANEMIN) flags="$flags -DRG_SWEEP16_LANEMIN";; STALL2) flags="$flags -DRG_SWEEP16_STALLSTAT=2";; BANDLOOPS) flags="$flags -DRG_BAND_SIMD_LOOPS";; NOAHEAD) flags="$flags -DRG_SWEEP16_PROFILE_AHEAD=0";; STALL3) flags="$flags -DRG_SWEEP16_STALLSTAT=3";; G32) flags="$flags -DRG_SWEEP16_GATHER32=1";; REVKRUN0) flags="$flags -DRG_SWEEP16_KRUN_REV=0 -DRG_SWEEP16_KRUN=0";; GR*) v2=${f#GR}; flags="$flags -DRG_GATHER_PER_MEMBER_ROW=${v2%%x*} -DRG_GATHER_PER_MEMBER_RUN=${v2##*x}";; GFWD) flags="$flags -DRG_SWEEP16_GATHER_FWD=1";; NOGATHER) flags="$flags -DRG_SWEEP16_GATHER=0";; NOPF) flags="$flags -DRG_SWEEP16_PF=0";; GNOPH1) flags="$flags -DRG_G_NOPH1";; GNOPH3) flags="$flags -DRG_G_NOPH3";; RSH*) flags="$flags -DRG_SWEEP16_RETIRE_SHIFT=${f#RSH}";; *) flags="$flags -DRG_SWEEP16_$f";; esac; done
    rm -rf /tmp/rgvar_$v; mkdir -p /tmp/rgvar_$v
    cp -r recgraph_amd/csrc /tmp/rgvar_$v/csrc; mkdir -p /tmp/rgvar_$v/include; cp include/recgraph_hip.h /tmp/rgvar_$v/include/
    mkdir -p /tmp/rgvar_$v/x; mv /tmp/rgvar_$v/csrc /tmp/rgvar_$v/x/csrc; mkdir -p /tmp/rgvar_$v/include
    ( cd /tmp/rgvar_$v/x/csrc && rm -rf build && make -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function $flags" > /dev/null 2>&1 ) || { echo "build $v failed"; exit 1; }
    cp /tmp/rgvar_$v/x/librecgraph_hip.so tools/build/librecgraph_hip_$v.so
    echo "built $v ($flags)"
  done
  exit 0
fi
# (RG_NO_SPEC=1: with parts compiled out the speculative bound fails its check and the reads run twice; SPEC=1 keeps the
#  speculative bound for variants whose results are valid)
for v in $VARIANTS; do
  RG_NO_SPEC=$([ "$SPEC" = 1 ] && echo 0 || echo 1) RG_LIB_PATH=$PWD/tools/build/librecgraph_hip_$v.so python3 bench.py --config ${CFG:-C5} --steps 4 --warmup 1 --no-cpu --no-strong --no-probe --handles 1 2>/dev/null |
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$v', 'fwd', k.get('k_sweep16_fwd'), 'rev', k.get('k_sweep16_rev'), 'step', d['ms_per_step'])"
done
