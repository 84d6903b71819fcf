# round 4: the chain split's model constant (SMM_SPLIT_NS) and the spacer, 40 steps per setting (cfg3 default bench, same box)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for ns in 170 140 120 100 80 60; do for sp in 15; do
  SMM_SPLIT_NS=$ns SMM_SPLIT_SPACER_US=$sp SMM_VERBOSE=0 timeout -k 10 200 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-predict-e2e --second-seed -1 --no-strong-leg 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); rf=r['roofline']
print('SMM_SPLIT_NS=$ns spacer $sp us: %.3f ms/step, launches per step %.1f, critical %.3f rest %.3f' % (r['ms_per_step'], rf.get('launches_per_step') or 0, rf.get('critical_launch_ms') or -1, rf.get('rest_launch_ms') or -1))"
done; done 2>&1 | tee gpurun_out/r4y.txt
