#!/bin/bash
# SQ counter pass of one bench workload for one engine build: tools/pmc_quick.sh <lib.so> [workload]   (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export BMO_ENGINE_LIB=$PWD/$1
W=${2:-c2}
T=$(basename $1 .so)_$W
O=gpurun_out/pmcq_$T; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/sq -- python3 bench.py --workload $W --steps 1 --warmup 0 --cpu-sample 0 --no-extras > /dev/null 2> $O/sq.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA --output-format csv -d $O/sq2 -- python3 bench.py --workload $W --steps 1 --warmup 0 --cpu-sample 0 --no-extras > /dev/null 2> $O/sq2.err
for d in sq sq2; do python3 tools/pmc_summary.py $O/$d | tail -1; done
find $O -name "*.db" -delete
