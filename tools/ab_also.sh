#!/bin/bash
# A/B on one box of one "also" line of bench.py (png, zlib6, mix, level1, rle, encode): tools/ab_also.sh <line> reps libA.so libB.so ...
line=$1; reps=$2; shift; shift
for r in $(seq 1 $reps); do
  for lib in "$@"; do
    FDH_LIB=$lib python bench.py --no-cpu-baseline --also-select $line --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for a in d.get('also', []):
    print('%-28s %s' % ('$lib', json.dumps({k: a[k] for k in a if k in ('metric','value','ms_per_step','ms','roofline')})[:300]))"
  done
done
