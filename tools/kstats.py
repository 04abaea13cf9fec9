#!/usr/bin/env python3
"""Average duration of the general encoder's kernels from a rocprofv3 --kernel-trace --stats --output-format csv
directory: python tools/kstats.py <dir>"""
import csv,glob,sys
for f in glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "parse" in r["Name"] or "write_kernel" in r["Name"]: print("  ", r["Name"][:60], r["Calls"], "%.2f ms" % (float(r["AverageNs"])/1e6))
