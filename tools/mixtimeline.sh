#!/bin/bash
# Timeline of the kernels of ONE call on the mixed batch (BASELINE config 5), from a kernel trace:
#   bash tools/mixtimeline.sh <tag>   (GPU box, via gpurun) -> gpurun_out/mixtl_<tag>.txt
TAG=${1:-mix}
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/mixtl_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --also-select mix > $OUT/log.txt 2>&1
python3 - <<PY > gpurun_out/mixtl_$TAG.txt
import csv, glob
rows = []
for f in glob.glob("$OUT/p/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", "")))
rows.sort()
# the last stream_order_kernel pair starts the last call
starts = [i for i, r in enumerate(rows) if "stream_order" in r[2]]
last = starts[-1]
while last > 0 and "stream_order" in rows[last - 1][2]:
    last -= 1
t0 = rows[last][0]
for s, e, n, q in rows[last:]:
    print("%9.3f ms  +%8.3f ms  q%s  %s" % ((s - t0) / 1e6, (e - s) / 1e6, q, n))
PY
cat gpurun_out/mixtl_$TAG.txt
rm -rf $OUT/p
