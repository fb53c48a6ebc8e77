#!/bin/bash
# UTCL1 (L1 TLB) counters of one bench workload: tools/pmc_tlb.sh <lib.so> [workload]   (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export BMO_ENGINE_LIB=$PWD/$1
W=${2:-c2}
O=gpurun_out/pmctlb_$(basename $1 .so)_$W; rm -rf $O; mkdir -p $O
rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum --output-format csv -d $O/a -- python3 bench.py --workload $W --steps 1 --warmup 0 --cpu-sample 0 --no-extras > /dev/null 2> $O/a.err
rocprofv3 --pmc TCP_UTCL1_SERIALIZATION_STALL TCP_UTCL1_THRASHING_STALL TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS --output-format csv -d $O/b -- python3 bench.py --workload $W --steps 1 --warmup 0 --cpu-sample 0 --no-extras > /dev/null 2> $O/b.err
for d in a b; do python3 tools/pmc_summary.py $O/$d | tail -4; done
find $O -name "*.db" -delete
