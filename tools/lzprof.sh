cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lzc -- python3 tools/gendiag.py 65536 0 > gpurun_out/prof_lzc.log 2>&1
python3 - <<'PY'
import csv,glob
for f in glob.glob("gpurun_out/prof_lzc/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fdh" in r["Name"]:
            print(r["Name"][:60], r["Calls"], "avg_ms", round(float(r["AverageNs"])/1e6,3), "total_ms", round(float(r["TotalDurationNs"])/1e6,2))
PY
rm -rf gpurun_out/prof_lzc
