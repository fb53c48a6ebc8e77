#!/bin/bash
# The alternative paths of the engine, each over the parity, fuzz, retrace and splitter-chain GPU tests (run on the GPU box):
#   tools/alt_paths.sh > gpurun_out/alt_paths.txt
T="tests/test_gpu_parity.py tests/test_fuzz.py tests/test_retrace.py tests/test_retrace_stale.py tests/test_splitter_chain.py"
for e in "BMO_FORCE_SORT_ORDER=1" "BMO_FORCE_DEEP_ORDER=1" "BMO_THIN_WAVES=1000000" "BMO_INWAVE_MAX=0" "BMO_FUSE=3 BMO_FUSE_GAUSS=2" "BMO_KEEP_KIDS=0" "BMO_LPT=0 BMO_REVERSE=0" \
         "BMO_REVERSE=2" "BMO_WIDE_MIN_WAVES=0" "BMO_ROOT_ORDER=chord" "BMO_ROOT_ORDER=none" "BMO_ROOT_ORDER=mask BMO_KEEP_KIDS=0 BMO_LPT=0 BMO_REVERSE=0"; do
  echo "== $e: $(env $e python -m pytest $T -m gpu -q 2>&1 | tail -1)"
done
