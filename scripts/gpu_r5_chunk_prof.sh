# round 5: kernel statistics of a time-split decode (cfg1, cfg3): how long the prefix sums, the units' DP, the stitch and the repair launch take
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${SMM_TAG:-r5_chunk}
for w in ${SMM_PROF_WL:-cfg1 cfg3}; do
  rm -rf gpurun_out/prof_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$w -- python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-predict-e2e --no-strong-leg --second-seed -1 > gpurun_out/prof_$w.log 2>&1
  grep "^{\"metric" gpurun_out/prof_$w.log | tail -1 | python -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$w', round(j['value']/1e6,1), 'M', round(j['ms_per_step'],3), 'ms/step; DP launches', r['launches_per_step'], 'mean', round(r['kernel_ms'],3), 'crit', r['critical_launch_ms'], 'rest', r['rest_launch_ms'], 'time_split', j.get('time_split'))"
  python - <<PY
import csv, glob
for f in glob.glob('gpurun_out/prof_$w/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'smm_' in r['Name']:
            print('%-90s calls %5s  avg %10.1f us  total %10.1f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3))
PY
done
