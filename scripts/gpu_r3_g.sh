set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_eval.py tests/test_gpu_model.py -q -x > gpurun_out/r3g_pytest.log 2>&1 ; echo "tests rc=$?"
tail -8 gpurun_out/r3g_pytest.log
timeout -k 10 300 python scripts/probe_eval.py 2>&1 | head -7
timeout -k 10 600 python bench.py --steps 5 --warmup 2 2>gpurun_out/r3g_cfg3.err | tail -1 > gpurun_out/r3g_cfg3.json
python - <<'PY'
import json
r = json.load(open('gpurun_out/r3g_cfg3.json'))
print(round(r['value']/1e6, 1), 'Mframes/s', round(r['ms_per_step'], 3), 'ms dp', round(r['roofline']['kernel_ms'], 3))
print('host_features', r.get('host_features'))
print('e2e', r.get('predict_end_to_end'))
print('parity', {k: v for k, v in r.get('parity', {}).items() if k != 'what'})
print('eval ms', r['evaluation']['ms'], 'cpu', r.get('cpu_baseline', {}).get('value'), r.get('cpu_factored', {}).get('value'), r.get('cpu_factored', {}).get('cores'))
PY
tail -5 gpurun_out/r3g_cfg3.err
