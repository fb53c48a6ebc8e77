#!/bin/bash
# Wave timelines of one workload with the developer build (build_ab/libbmo_tl.so: -DBMO_DEV_RAY_LDS_ONLY -DBMO_DEV_TIMELINE), natural and reversed
# workgroup order:  tools/timeline.sh c2s   (run on the GPU box; output under gpurun_out/timeline_<workload>_rev<0|1>.txt)
w=${1:-c2s}
mkdir -p gpurun_out
for rev in ${REVS:-0 1 2}; do
  BMO_REVERSE=$rev BMO_TIMELINE=1 BMO_ENGINE_LIB=$PWD/build_ab/libbmo_tl.so python tools/step_times.py $w 2>&1 | grep -v "^\[bmo\] \(phase\|step .* launching\|roots\|init\|retrace\)" > gpurun_out/timeline_${w}_rev$rev.txt
done
