#!/bin/bash
# A/B of engine builds over bundle sizes: tools/ab_sizes.sh <workload> "<sizes>" lib1 lib2 ...   (kernel ms per solve, best of REPS runs of 4 solves)
wl=$1; sizes=$2; shift 2
for n in $sizes; do
  for lib in "$@"; do
    best=""
    for rep in $(seq 1 ${REPS:-2}); do
      v=$(BMO_ENGINE_LIB=$PWD/$lib python bench.py --workload $wl --rays $n --steps 4 --warmup 1 --cpu-sample 0 --no-extras 2>gpurun_out/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.4f %.4f' % (d['roofline']['kernel_ms_per_solve'], d['ms_per_step']))")
      best="$best | $v"
    done
    printf "%-4s rays %8d  %-30s kernel ms / ms per step: %s\n" $wl $n $lib "$best"
  done
done
