#!/bin/bash
# Per-kernel times of the mixed batch (BASELINE config 5): bash tools/mixprof.sh <tag>   (GPU box, via gpurun)
TAG=${1:-mix}
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/mixprof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --also-select mix > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/p/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "inflate" in r["Name"] or "order" in r["Name"]:
            print("%-44s calls %3s avg %9.3f ms  max %9.3f ms" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
PY
tail -1 $OUT/log.txt | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print([(a['value'],a['ms_per_step']) for a in j['also']])"
rm -rf $OUT/p
