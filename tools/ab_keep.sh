for v in "" "BMO_KEEP_HI=0.0" "BMO_KEEP_HI=0.5 BMO_KEEP_LO=0.0 BMO_FUSE_MID=32" "BMO_KEEP_HI=0.9 BMO_KEEP_LO=0.3 BMO_FUSE_MID=8 BMO_FUSE_LO=4" "BMO_KEEP_HI=0.9 BMO_KEEP_LO=0.0 BMO_FUSE_MID=4"; do
 for wl in c2v c5; do
  env $v BMO_ENGINE_LIB=$PWD/build_ab/keep.so python bench.py --workload $wl --steps 2 --warmup 1 --cpu-sample 0 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-70s %-4s ms/step %7.3f  kernel avg %.3f ms x %d launches' % ('$v', '$wl', d['ms_per_step'], r['avg_launch_ms'], r['launches_per_step']))"
 done
done
