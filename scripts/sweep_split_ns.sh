# round 4: choose_split's slack model (SMM_SPLIT_NS: modelled DP time per frame, SMM_SPLIT_MARGIN) re-scanned on one box after the
# DP kernel went from ~250 to ~165 ns per frame: bench.py's timed step on cfg3 (seeds 2 and 1000) and cfg5 x 0.13 (one rank's share of 8)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/split_ns.txt
for seed in 2 1000; do
for ns in 100 130 165 200 250 330 450; do
  SMM_SPLIT_NS=$ns timeout -k 10 120 python bench.py --steps 20 --warmup 3 --seed $seed --no-cpu-baseline --no-predict-e2e --no-strong-leg --second-seed -1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('seed $seed SMM_SPLIT_NS=$ns: step %.3f ms  critical %.3f  rest %.3f' % (d['ms_per_step'], r.get('critical_launch_ms') or 0, r.get('rest_launch_ms') or 0))" >> gpurun_out/split_ns.txt || exit 1
done; done
cat gpurun_out/split_ns.txt
