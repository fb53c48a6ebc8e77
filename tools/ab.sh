#!/bin/bash
# A/B of engine builds on one GPU box: tools/ab.sh <rays> <steps> lib1 lib2 ...
rays=$1; steps=$2; shift 2
mkdir -p gpurun_out
for lib in "$@"; do
  for rep in 1 2; do
    BMO_ENGINE_LIB=$PWD/$lib python bench.py --rays $rays --steps $steps --warmup 1 --cpu-sample 0 2>gpurun_out/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-28s value %.3e  ms/step %.2f  kernel avg %.3f ms x %d launches' % ('$lib', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['launches_per_step']))"
  done
done
