set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r2g_pytest_all.log 2>&1 ; echo "all tests rc=$?"
tail -4 gpurun_out/r2g_pytest_all.log
timeout -k 10 600 python bench.py --steps 5 --warmup 2 2> gpurun_out/r2g_bench_cfg3.err | tail -1 > gpurun_out/r2g_bench_cfg3.json
timeout -k 10 600 python bench.py --workload cfg4 --steps 5 --warmup 2 2> gpurun_out/r2g_bench_cfg4.err | tail -1 > gpurun_out/r2g_bench_cfg4.json
python - <<'PY'
import json
r=json.load(open('gpurun_out/r2g_bench_cfg3.json'))
print('cfg3', round(r['value']/1e6,1), 'Mframes/s', round(r['ms_per_step'],3), 'ms/step dp_ms', round(r['roofline']['kernel_ms'],3))
print('fit', r['fit_stats']['ms'], r['fit_stats']['roofline']['frac'])
print('e2e', {k: (round(v['ms'],2) if isinstance(v, dict) else '') for k, v in r.get('predict_end_to_end').items()})
print('cpu', r['cpu_baseline']['value'], r['cpu_factored']['value'], r['cpu_factored']['cores'])
r=json.load(open('gpurun_out/r2g_bench_cfg4.json'))
l=r['logz_fwd_bwd']
print('cfg4 decode', round(r['value']/1e6,1), 'logz packed', round(l['packed']['value']/1e6,2), 'M f/s', round(l['packed']['ms'],2), 'ms; per_batch', round(l['per_batch']['value']/1e6,2), 'kernels', l['kernels']['logz_fwd_ms'], l['kernels']['logz_bwd_ms'], 'cpu', l['cpu_baseline'])
PY
