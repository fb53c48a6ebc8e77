#!/bin/bash
# Profile set of one round for EVERY bench workload: rocprofv3 kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes
# (separate --pmc runs, never combined with a trace domain) per workload, SQ counter passes and the VALU mix for the headline one.
#   tools/profile_round.sh [workloads...]      (run on the GPU box; results under gpurun_out/prof_round/, collected by
#   tools/collect_profiles.py <tag> into profiles/)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
WL=${@:-c2s c2 c2v c3 c4 c5}
O=gpurun_out/prof_round; rm -rf $O; mkdir -p $O
for w in $WL; do
  mkdir -p $O/$w
  B="python3 bench.py --workload $w --cpu-sample 0 --no-extras"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w/stats -- $B --steps 5 --warmup 1 > $O/$w/bench_under_rocprof.json 2> $O/$w/stats.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$w/fetch -- $B --steps 1 --warmup 0 > /dev/null 2> $O/$w/fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/$w/write -- $B --steps 1 --warmup 0 > /dev/null 2> $O/$w/write.err
  if true; then  # SQ + VALU-mix passes for every workload (the ragged one too since the end of round 4: its lane utilisation is the point)
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/$w/sq -- $B --steps 1 --warmup 0 > /dev/null 2> $O/$w/sq.err
    rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH --output-format csv -d $O/$w/sq2 -- $B --steps 1 --warmup 0 > /dev/null 2> $O/$w/sq2.err
    rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU --output-format csv -d $O/$w/sq3 -- $B --steps 1 --warmup 0 > /dev/null 2> $O/$w/sq3.err
  fi
  for d in fetch write sq sq2 sq3; do [ -d $O/$w/$d ] && python3 tools/pmc_summary.py $O/$w/$d > $O/$w/${d}_summary.txt 2>&1; done
  find $O/$w -name "*kernel_stats.csv" -exec cp {} $O/$w/kernel_stats.csv \;
  echo "profiled $w"
done
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
