set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_module.py tests/test_gpu_fullsize.py tests/test_gpu_eval.py tests/test_gpu_fit.py -x -q -m gpu -k "emission or cfg or golden or reference or eval or fit or accuracy" 2>&1 | tail -3
rm -rf gpurun_out/prof_em
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_em -- python scripts/probe_emission.py cfg3 > gpurun_out/r2e_probe.txt 2>&1
tail -3 gpurun_out/r2e_probe.txt
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_em/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'smm_' in r['Name']:
        print(r['Name'][:70], 'calls', r['Calls'], 'avg_us', round(float(r['AverageNs'])/1e3, 1), 'min_us', round(float(r['MinNs'])/1e3, 1))
PY
