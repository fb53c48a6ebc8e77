#!/bin/bash
# HBM traffic counters (FETCH_SIZE, WRITE_SIZE: separate passes) of one bench workload for one engine build:
# tools/pmc_traffic.sh <lib.so> [workload]   (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export BMO_ENGINE_LIB=$PWD/$1
W=${2:-c2}
O=gpurun_out/pmctr_$(basename $1 .so)_$W; rm -rf $O; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/$c -- python3 bench.py --workload $W --steps 1 --warmup 0 --cpu-sample 0 --no-extras > /dev/null 2> $O/$c.err
  python3 tools/pmc_summary.py $O/$c | tail -5 | head -4
done
find $O -name "*.db" -delete
