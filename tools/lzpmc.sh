#!/bin/bash
# PMC counters of the LZ-window kernel on the zlib level-6 streams of the bench data (tools/gendiag.py workload), two passes
# (counters never combined with tracing): usage (via gpurun): bash tools/lzpmc.sh <tag> [n_streams] [lib]
TAG=$1; N=${2:-16384}; LIB=${3:-}
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
[ -n "$LIB" ] && export FDH_LIB=$PWD/$LIB
OUT=gpurun_out/lzpmc_$TAG
rm -rf $OUT; mkdir -p $OUT
# (the program itself behind `--`; the watchdog goes outside)
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/p1 -- python3 tools/gendiag.py $N 0 > $OUT/log1.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/p2 -- python3 tools/gendiag.py $N 0 > $OUT/log2.txt 2>&1
python3 - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if "inflate_lz" in k: print("$TAG", k, "per stream:", {c: round(sum(x)/len(x)/$N,1) for c,x in v.items()})
PY
tail -3 $OUT/log1.txt
rm -rf $OUT/p1 $OUT/p2
