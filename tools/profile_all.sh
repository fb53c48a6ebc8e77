#!/bin/bash
# Full profile set of the default bench (config C2): kernel stats + SQ / FETCH / WRITE counter passes.  Run on the GPU box.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-extras > $O/bench_stats.json 2> $O/stats.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/sq -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-extras > /dev/null 2> $O/sq.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d $O/sq2 -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-extras > /dev/null 2> $O/sq2.err
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU --output-format csv -d $O/sq3 -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-extras > /dev/null 2> $O/sq3.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-extras > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-extras > /dev/null 2> $O/write.err
for d in sq sq2 sq3 fetch write; do python3 tools/pmc_summary.py $O/$d > $O/${d}_summary.txt 2>&1 || true; done
find $O -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
ls $O
