#!/bin/bash
# Profiles bench.py on the GPU box with rocprofv3: one kernel-trace/stats run, then separate
# --pmc passes (never combined with tracing, see the task notes).  Usage (via gpurun):
#   bash tools/profile.sh <tag> [bench args...]
# Raw output goes to gpurun_out/prof_<tag>/; tools/summarize_profile.py turns it into
# profiles/<tag>_*.csv / .json.
set -u
TAG=${1:-r01}; shift || true
ARGS=${@:---steps 20 --warmup 5 --no-cpu-baseline --no-also}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1
for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  NAME=$(echo $PASS | cut -d' ' -f1)
  rocprofv3 --pmc $PASS --output-format csv -d $OUT/pmc_$NAME -- python3 bench.py $ARGS > $OUT/bench_pmc_$NAME.log 2>&1
done
python3 tools/summarize_profile.py $OUT $TAG
# keep only the condensed summary (raw traces are tens of MiB)
rm -rf $OUT/trace $OUT/pmc_*
