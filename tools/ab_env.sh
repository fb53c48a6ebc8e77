#!/bin/bash
# A/B of environment switches on one build, interleaved over two rounds: tools/ab_env.sh <lib> <workloads...> -- "VAR=a" "VAR=b VAR2=c" ...
lib=$1; shift
wls=()
while [ "$1" != "--" ]; do wls+=("$1"); shift; done
shift
for rep in 1 2; do
  for wl in "${wls[@]}"; do
    for e in "$@"; do
      env $e BMO_ENGINE_LIB=$PWD/$lib python bench.py --workload $wl --steps 5 --warmup 1 --cpu-sample 0 --no-extras 2>>gpurun_out/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-44s %-4s ms/step %7.3f  kernel/solve %.3f ms, %d launches' % ('$e', '$wl', d['ms_per_step'], r['kernel_ms_per_solve'], r['launches_per_step']))"
    done
  done
done
