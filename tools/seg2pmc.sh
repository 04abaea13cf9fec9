#!/bin/bash
# PMC counters of the interval kernel alone (tools/seg2diag.py workload): usage: bash tools/seg2pmc.sh <tag> [lib]
TAG=$1; LIB=${2:-}
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
[ -n "$LIB" ] && export FDH_LIB=$PWD/$LIB
OUT=gpurun_out/s2pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
# (the program itself behind `--`: with --pmc the profiler's library initialises the GPU before the program starts, and
#  a launcher in between -- timeout, env, bash -c -- would exec from a process that has touched the GPU; the watchdog goes outside)
timeout 180 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/p -- python3 tools/seg2diag.py 65536 D > $OUT/log.txt 2>&1
python3 - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(list))
for f in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if "seg2" in k: print("$TAG", k, {c: round(sum(x)/len(x)/65536,1) for c,x in v.items()})
PY
grep "interval kernel alone" $OUT/log.txt
rm -rf $OUT/p
