#!/bin/bash
# Every profile the docs quote, one after the other (GPU box, via gpurun): kernel stats + PMC of
#   <round>          the headline decode (bench.py defaults)
#   <round>enc       the ultra-fast encoder
#   <round>png       decode + PNG reconstruction        (--also-select png)
#   <round>level1 / <round>rle   the general encoder   (--also-select level1 / rle)
#   <round>zlib6     decode of zlib level-6 streams     (--format zlib6)
#   <round>mix       the mixed batch (BASELINE config 5) (--also-select mix)
# and merges the per-path HBM traffic into gpurun_out/traffic_merged.json (copy it to
# profiles/traffic_latest.json together with gpurun_out/prof_*/summary/*.csv / *_pmc.json).
R=${1:-r04}
cd "$GRAFT_REPO_ROOT"
bash tools/profile.sh $R --steps 20 --warmup 5 --no-cpu-baseline --no-also
bash tools/profile.sh ${R}enc --mode encode --steps 10 --warmup 3 --no-cpu-baseline --no-also
FDH_PROFILE_KEY=png bash tools/profile.sh ${R}png --steps 4 --warmup 1 --no-cpu-baseline --also-select png
FDH_PROFILE_KEY=level1 bash tools/profile.sh ${R}level1 --steps 4 --warmup 1 --no-cpu-baseline --also-select level1
FDH_PROFILE_KEY=rle bash tools/profile.sh ${R}rle --steps 4 --warmup 1 --no-cpu-baseline --also-select rle
FDH_PROFILE_KEY=zlib6 bash tools/profile.sh ${R}zlib6 --format zlib6 --steps 5 --warmup 1 --no-cpu-baseline --no-also
FDH_PROFILE_KEY=mix bash tools/profile.sh ${R}mix --steps 4 --warmup 1 --no-cpu-baseline --also-select mix
python3 - <<PY
import json, glob
out, sha = {}, None
for f in sorted(glob.glob("gpurun_out/prof_${R}*/summary/traffic_latest.json")):
    d = json.load(open(f))
    s = d.pop("kernel_source_sha", None)
    if sha is None:
        sha = s
    if s != sha:
        print("skipping", f, "(other kernel sources)")
        continue
    for k, v in d.items():
        out.setdefault(k, v)
out["kernel_source_sha"] = sha
json.dump(out, open("gpurun_out/traffic_merged.json", "w"), indent=1)
print(json.dumps({k: (v.get("hbm_bytes_per_step") if isinstance(v, dict) else v) for k, v in out.items()}))
PY
