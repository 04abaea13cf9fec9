#!/bin/bash
# A/B on one box: the headline bench line for every library given (FDH_LIB), alternating, `reps` times.
#   tools/ab.sh reps libA.so libB.so ...   (on the GPU box; prints ms_per_step / kernel min per library and repetition)
reps=$1; shift
for r in $(seq 1 $reps); do
  for lib in "$@"; do
    FDH_LIB=$lib python bench.py --no-also --no-cpu-baseline --steps 40 ${AB_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-50s ms %.4f  kernel avg %.4f min %.4f' % ('$lib', d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['kernel_ms_min']))"
  done
done
