#!/bin/bash
# Per-kernel times of single entries of the mix pool: bash tools/mixslow_prof.sh <entry[,entry]> [flags]
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/mixslow_prof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -- python3 tools/mixslow.py ${2:-0} $1 > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/p/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "inflate" in r["Name"]:
            print("%-44s calls %3s avg %9.3f ms  max %9.3f ms" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
PY
grep " ms " $OUT/log.txt
rm -rf $OUT/p
