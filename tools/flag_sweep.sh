#!/bin/bash
wl=$1; steps=$2; shift 2
for rep in 1 2; do
  for lib in "$@"; do
    BMO_ENGINE_LIB=$PWD/build_ab/libbmo_$lib.so timeout -k 10 120 python bench.py --workload $wl --steps $steps --warmup 1 --cpu-sample 0 --no-extras 2>gpurun_out/ab.err | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']
    print('%-10s %-4s ms/step %7.3f  kernel avg %.3f ms x %d' % ('$lib', '$wl', d['ms_per_step'], r['avg_launch_ms'], r['launches_per_step']))
except Exception as e: print('$lib', 'FAILED', e)"
  done
done
