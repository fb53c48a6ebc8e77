#!/bin/bash
# A/B of engine builds on one GPU box, interleaved: tools/ab.sh <workload> <steps> lib1 lib2 ...   (libs relative to the repo root)
wl=$1; steps=$2; shift 2
mkdir -p gpurun_out
for rep in $(seq 1 ${REPS:-3}); do
  for lib in "$@"; do
    BMO_ENGINE_LIB=$PWD/$lib python bench.py --workload $wl --steps $steps --warmup 1 --cpu-sample 0 --no-extras 2>gpurun_out/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-34s %-4s value %.3e  ms/step %7.3f  kernel/solve %.3f ms, %d launches' % ('$lib', '$wl', d['value'], d['ms_per_step'], r['kernel_ms_per_solve'], r['launches_per_step']))"
  done
done
