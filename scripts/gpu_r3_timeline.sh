set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-predict-e2e --second-seed -1 > gpurun_out/tl.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/tl/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'smm_' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last full decode step: find the last 2 smm_viterbi band launches
idx = [i for i, r in enumerate(rows) if 'smm_viterbi_kernel' in r['Kernel_Name']]
# print a window of launches around the 3rd-last .. last viterbi
lo = max(0, idx[-6] - 8)
t0 = int(rows[lo]['Start_Timestamp'])
for r in rows[lo:idx[-1] + 1]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print('%9.1f us -> %9.1f us  (%8.1f us)  q=%s  grid=%s  %s' % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get('Queue_Id'), r.get('Grid_Size_X', r.get('Grid_Size')), r['Kernel_Name'][:60]))
PY
