# rocprofv3 kernel stats of the default bench (cfg3 seed 2) + PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_cfg3 gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg3 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-predict-e2e > gpurun_out/prof_cfg3.log 2>&1
tail -1 gpurun_out/prof_cfg3.log | cut -c1-400
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_cfg3/**/*kernel_stats.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:14]:
    print(r['Name'][:80], '| calls', r['Calls'], '| avg_us', round(float(r['AverageNs'])/1e3, 1), '| total_ms', round(float(r['TotalDurationNs'])/1e6, 2))
PY
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-predict-e2e > gpurun_out/pmc_$c.log 2>&1
done
python scripts/pmc_summary.py gpurun_out cfg3 | tail -30
