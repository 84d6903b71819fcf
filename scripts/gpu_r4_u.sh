# round 4: kernel timeline of the default bench's steps (rocprofv3 --kernel-trace): what the step is made of beside the DP launch
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/tl4
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl4 -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-predict-e2e --second-seed -1 --no-strong-leg > gpurun_out/tl4.log 2>&1
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob('gpurun_out/tl4/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:48], r.get('Stream_Id', r.get('Queue_Id', '?'))))
rows.sort()
# the last two steps: find the last 2 occurrences of the band tables kernel
idx = [i for i, r in enumerate(rows) if 'band_tables' in r[2]]
start = idx[-2] - 3 if len(idx) >= 2 else 0
t0 = rows[start][0]
for s, e, n, q in rows[start:]:
    print('%9.1f us  +%8.1f us  q%-3s %s' % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
PY
