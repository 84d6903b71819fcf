# SQ counters of our kernels on the default bench workload (separate pass: --pmc with --kernel-trace only)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_sq
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc_sq -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-predict-e2e --no-strong-leg --second-seed -1 > gpurun_out/pmc_sq.log 2>&1
python - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob('gpurun_out/pmc_sq/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        if 'smm_' not in k: continue
        rows[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVES': cnt[k] += 1
for k, d in rows.items():
    n = cnt[k] or 1
    print(k, 'launches', n, {c: round(v / n) for c, v in d.items()})
PY
