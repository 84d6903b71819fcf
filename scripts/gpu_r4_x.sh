# round 4: the Viterbi GPU tests through the shipped library, the default bench (40 steps), and the step's kernel timeline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then timeout -k 10 700 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_graph.py tests/test_gpu_model.py -m gpu -q -x > gpurun_out/r4x_tests.txt 2>&1; rc=$?; else rc=0; fi
tail -3 gpurun_out/r4x_tests.txt; echo "tests rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
for i in 1 2; do timeout -k 10 200 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-predict-e2e --second-seed -1 --no-strong-leg 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); rf=r['roofline']
print('%.3f ms/step, %.1f M frames/s, critical %.3f rest %.3f' % (r['ms_per_step'], r['value']/1e6, rf.get('critical_launch_ms') or -1, rf.get('rest_launch_ms') or -1))"; done
rm -rf gpurun_out/tl5
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl5 -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-predict-e2e --second-seed -1 --no-strong-leg > gpurun_out/tl5.log 2>&1
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob('gpurun_out/tl5/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60], r.get('Stream_Id', r.get('Queue_Id', '?'))))
rows.sort()
idx = [i for i, r in enumerate(rows) if 'viterbi_kernel' in r[2]]
a = idx[10] - 5; b = idx[12] + 3
t0 = rows[a][0]
for s, e, n, q in rows[a:b]:
    print('%9.1f us  +%8.1f us  q%-3s %s' % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
PY
