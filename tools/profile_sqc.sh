# Scalar data cache / instruction cache counters of one solve of c2s and c5 (run on the GPU box): tools/profile_sqc.sh ; SQC_SET="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" tools/profile_sqc.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/sqc; rm -rf $O; mkdir -p $O
for w in c2s c5; do
  rocprofv3 --pmc ${SQC_SET:-SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE} --output-format csv -d $O/$w -- python3 bench.py --workload $w --cpu-sample 0 --no-extras --steps 1 --warmup 0 > /dev/null 2> $O/$w.err
  python3 tools/pmc_summary.py $O/$w > $O/${w}_summary.txt 2>&1; tail -3 $O/${w}_summary.txt
done
rocprofv3 --list-avail 2>/dev/null | grep -o "SQC_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $O/avail.txt; cat $O/avail.txt
