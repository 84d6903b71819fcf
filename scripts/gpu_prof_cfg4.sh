cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg4 -- python bench.py --workload cfg4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_cfg4.log 2>&1
f=$(ls -t $(find gpurun_out/prof_cfg4 -name "*kernel_stats.csv") | head -1)
python - "$f" <<'PY'
import csv, sys
rows=list(csv.reader(open(sys.argv[1])))
for r in rows[1:14]:
    print(r[0][:70].replace('\n',' '), r[1], 'avg_us', round(float(r[3])/1e3,1), 'pct', r[4])
PY
