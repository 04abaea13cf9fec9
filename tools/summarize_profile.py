#!/usr/bin/env python3
"""Condenses a tools/profile.sh run into small files under gpurun_out/prof_<tag>/summary/ (copy
them into profiles/): kernel stats of our kernels, per-kernel PMC averages, and the HBM traffic
per launch with the gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE counts 64 B per 128-B
request for wide streaming reads -> doubled; both counters are in KiB)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
sumdir = os.path.join(out, "summary")
os.makedirs(sumdir, exist_ok=True)
OURS = ("fdh::",)


def ours(name):
    return any(k in name for k in OURS)


# kernel stats
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if ours(r["Name"])]
    with open(os.path.join(sumdir, "%s_kernel_stats.csv" % tag), "w") as g:
        w = csv.writer(g)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]])
    for r in rows:
        print("stats", r["Name"][:60], "calls", r["Calls"], "avg_ms", float(r["AverageNs"]) / 1e6)

# pmc
pmc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if ours(name):
            pmc[name.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in pmc.items():
    res[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    res[k]["_launches"] = {c: len(v) for c, v in cs.items()}
traffic = {}
for k, cs in res.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        fetch_b = cs["FETCH_SIZE"] * 1024 * 2   # gfx950: wide streaming reads are under-counted 2x
        write_b = cs["WRITE_SIZE"] * 1024
        traffic[k] = {"fetch_bytes_corrected": fetch_b, "fetch_bytes_raw": cs["FETCH_SIZE"] * 1024,
                      "write_bytes": write_b, "hbm_bytes_per_launch": fetch_b + write_b}
json.dump({"pmc_avg_per_launch": res, "traffic": traffic}, open(os.path.join(sumdir, "%s_pmc.json" % tag), "w"), indent=1)
# per-step HBM traffic of the decode / encode paths (all kernels of the path added up)
step = {}
dec = sum(t["hbm_bytes_per_launch"] for k, t in traffic.items() if "inflate" in k)
enc = sum(t["hbm_bytes_per_launch"] for k, t in traffic.items() if "deflate" in k)
if dec:
    step["decode"] = {"hbm_bytes_per_step": dec, "kernels": [k for k in traffic if "inflate" in k],
                      "note": "FETCH_SIZE*1024*2 (gfx950 under-count of wide reads) + WRITE_SIZE*1024, separate --pmc passes"}
if enc:
    step["encode"] = {"hbm_bytes_per_step": enc, "kernels": [k for k in traffic if "deflate" in k],
                      "note": "FETCH_SIZE*1024*2 + WRITE_SIZE*1024, separate --pmc passes"}
# the "also" lines of bench.py (profiled one by one: tools/profile.sh <tag> --also-select <name> ...)
# (only the kernels of the timed step: png_wave_kernel is a one-off property-check launch of bench.py, not part of it)
png = sum(t["hbm_bytes_per_launch"] for k, t in traffic.items() if "png_pipe" in k)
fenc = sum(t["hbm_bytes_per_launch"] for k, t in traffic.items() if "deflate_ultrafast_kernel_t<true>" in k or "deflate_ultrafast_kernel_t<(bool)1>" in k)
gen = sum(t["hbm_bytes_per_launch"] for k, t in traffic.items() if "deflate_parse" in k or "deflate_write" in k)
also_key = os.environ.get("FDH_PROFILE_KEY", "")
if also_key == "png" and png:
    step = {"png": {"hbm_bytes_per_step": png + dec, "kernels": [k for k in traffic if "png_pipe" in k or "inflate" in k],
                    "note": "decode kernels + reconstruction kernel of one fdh_inflate_png_batch call"}}
    if fenc:
        step["filterenc"] = {"hbm_bytes_per_step": fenc, "kernels": [k for k in traffic if "deflate_ultrafast_kernel_t" in k and "true" in k or "(bool)1" in k],
                             "note": "the fused filter + ultra-fast encode kernel of one fdh_png_filter_deflate_ultrafast_batch call"}
elif also_key in ("level1", "rle") and gen:
    step = {also_key: {"hbm_bytes_per_step": gen, "kernels": [k for k in traffic if "deflate_parse" in k or "deflate_write" in k],
                       "note": "parser + block writer of one fdh_deflate_general_batch call"}}
elif also_key == "mix" and dec:
    step = {"mix": {"hbm_bytes_per_step": dec, "kernels": step["decode"]["kernels"],
                    "note": "all decode kernels of one fdh_inflate_batch call on the mixed batch"}}
elif also_key == "zlib6" and dec:
    step = {"zlib6": {"hbm_bytes_per_step": dec, "kernels": step["decode"]["kernels"],
                      "note": "all decode kernels of one fdh_inflate_batch call on zlib level-6 streams"}}
# bench.py quotes these figures only when they were measured on exactly this kernel source
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import bench
    step["kernel_source_sha"] = bench.kernel_source_sha()
except Exception:
    pass
json.dump(step, open(os.path.join(sumdir, "traffic_latest.json"), "w"), indent=1)
for k, cs in res.items():
    print("pmc", k)
    for c, v in sorted(cs.items()):
        if not c.startswith("_"):
            print("    %-24s %.4g" % (c, v))
for k, t in traffic.items():
    print("traffic", k, {a: "%.4g" % b for a, b in t.items()})
