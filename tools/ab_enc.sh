#!/bin/bash
# A/B on one box, encoder line: tools/ab_enc.sh reps libA.so libB.so ...
reps=$1; shift
for r in $(seq 1 $reps); do
  for lib in "$@"; do
    FDH_LIB=$lib python bench.py --mode encode --no-also --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s ms %.4f  kernel avg %.4f min %.4f frac %.4f' % ('$lib', d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['kernel_ms_min'], d['roofline']['frac']))"
  done
done
