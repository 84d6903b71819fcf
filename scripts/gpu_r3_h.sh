set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_model.py tests/test_gpu_module.py tests/test_gpu_random_sweeps.py -q -x > gpurun_out/r3h_pytest.log 2>&1 ; echo "tests rc=$?"
tail -8 gpurun_out/r3h_pytest.log
timeout -k 10 200 python scripts/probe_small.py > gpurun_out/r3h_probe_small.txt 2>&1; echo rc=$?
SMM_NO_BT_WINDOW=1 timeout -k 10 200 python scripts/probe_small.py > gpurun_out/r3h_probe_small_general.txt 2>&1; echo rc=$?
echo WINDOW; cat gpurun_out/r3h_probe_small.txt; echo GENERAL; cat gpurun_out/r3h_probe_small_general.txt
timeout -k 10 300 python bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline --no-predict-e2e 2>gpurun_out/r3h_cfg4.err | tail -1 > gpurun_out/r3h_cfg4.json
SMM_NO_BT_WINDOW=1 timeout -k 10 300 python bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline --no-predict-e2e 2>gpurun_out/r3h_cfg4g.err | tail -1 > gpurun_out/r3h_cfg4_general.json
python - <<'PY'
import json
for w in ('cfg4', 'cfg4_general'):
    r = json.load(open('gpurun_out/r3h_%s.json' % w))
    print(w, round(r['value']/1e6, 1), 'Mframes/s', round(r['ms_per_step'], 3), 'ms dp', round(r['roofline']['kernel_ms'], 3), 'mof', r['mof'])
PY
