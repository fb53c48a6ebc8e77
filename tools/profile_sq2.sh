#!/bin/bash
# Where do waves wait?  LDS / VMEM / SMEM latency levels, instruction fetch, VALU mix.  tools/profile_sq2.sh  (counter passes only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_sq2; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-extras > /dev/null 2> $O/p$i.err
  python3 tools/pmc_summary.py $O/p$i | tail -2
done
