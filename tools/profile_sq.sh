#!/bin/bash
# SQ counter passes only (fast): tools/profile_sq.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_sq; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/sq -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2> $O/sq.err
rocprofv3 --pmc SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $O/sq2 -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2> $O/sq2.err
for d in sq sq2; do python3 tools/pmc_summary.py $O/$d | tail -2; done
