#!/bin/bash
# rocprofv3 evidence for the round: kernel-trace stats + separate PMC passes (never combined with
# tracing domains). Run on the GPU box from the repo root: bash scripts/profile.sh <tag> [bench args...]
# BGTH_DEBUG_SKIP / BGTH_DEBUG_TIMES only act on the profiling build: make -C bgt_amd/csrc ABLATE=1 and
# BGT_AMD_LIB=$R/bgt_amd/lib/libbgt_hip_ablate.so
set -u
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT                                  # (a second run into the same tag must not leave the files of the first beside its own)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-secondary $*"
# the kernel trace runs as many steps as a default bench.py run warms up and times (the first launches of a process run 2-5 % slower:
# four calls averaged 10.34 ms for C2's kernel where bench.py's thirty measured 10.00); the counter passes serialise kernels and stay short
TRACE_STEPS=${TRACE_STEPS:-20}
echo "== kernel trace" 
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py --steps $TRACE_STEPS --warmup 5 --cpu-sample 0 --no-secondary --no-counters $* > $OUT/trace.log 2>&1
if [ -n "${TRACE_ONLY:-}" ]; then exit 0; fi
for PASS in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  N=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  echo "== pmc $PASS"
  rocprofv3 --pmc $PASS --output-format csv -d $OUT/pmc_$N -- $BENCH > $OUT/pmc_$N.log 2>&1
done
find $OUT -name "*.csv" | head -50
# FETCH_SIZE calibration on 1 GiB streamed with 4-byte and 16-byte loads
for W in 4 16; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib_w$W -- python $R/scripts/calib_fetch.py $W > $OUT/calib_w$W.log 2>&1
done
