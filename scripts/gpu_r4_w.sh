# round 4: choose_split's threshold against the round-4 DP kernel, 40 steps per setting, two passes (cfg3 default bench, same box)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for pass in 1 2; do for ns in 330 260 220 190 165 140 120; do
  SMM_SPLIT_NS=$ns SMM_SPLIT_MARGIN=400 timeout -k 10 200 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-predict-e2e --second-seed -1 --no-strong-leg 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); rf=r['roofline']
print('pass $pass SMM_SPLIT_NS=$ns: %.3f ms/step, critical %.3f rest %.3f' % (r['ms_per_step'], rf.get('critical_launch_ms') or -1, rf.get('rest_launch_ms') or -1))"
done; done 2>&1 | tee gpurun_out/r4w.txt
