# round 4: choose_split's model constants against the round-4 DP kernel (cfg3 default bench, same box)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for ns in 330 260 220 190 400; do for mg in 400 0; do
  SMM_SPLIT_NS=$ns SMM_SPLIT_MARGIN=$mg timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-predict-e2e --second-seed -1 --no-strong-leg 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); rf=r['roofline']
print('SMM_SPLIT_NS=$ns SMM_SPLIT_MARGIN=$mg: %.3f ms/step, critical %.3f rest %s' % (r['ms_per_step'], rf.get('critical_launch_ms') or -1, rf.get('rest_launch_ms')))"
done; done 2>&1 | tee gpurun_out/r4v.txt
SMM_NO_SPLIT=1 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-predict-e2e --second-seed -1 --no-strong-leg 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('SMM_NO_SPLIT: %.3f ms/step' % r['ms_per_step'])" | tee -a gpurun_out/r4v.txt
