# round 5: the time-split decode on the GPU -- its tests, the whole suite, and cfg1 / cfg3 with and without it on ONE box
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${SMM_TAG:-r5_chunk}
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/${T}_pytest.log 2>&1 ; echo "tests rc=$?"
tail -6 gpurun_out/${T}_pytest.log
for w in cfg1 cfg3; do
  for c in 0 1 0 1; do
    SMM_CHUNK=$c timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-predict-e2e --no-strong-leg --second-seed -1 2>gpurun_out/${T}_$w.err | tail -1 | python -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$w SMM_CHUNK=$c', round(j['value']/1e6,1), 'M', round(j['ms_per_step'],3), 'ms/step; DP launches', r['launches_per_step'], 'mean', round(r['kernel_ms'],3), 'crit', r['critical_launch_ms'], 'rest', r['rest_launch_ms'], 'time_split', j.get('time_split'))"
  done
done
