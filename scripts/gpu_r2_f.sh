set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_fit.py tests/test_gpu_model.py tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/r2f_bench_cfg3.err | tail -1 > gpurun_out/r2f_bench_cfg3.json
python - <<'PY'
import json
r=json.load(open('gpurun_out/r2f_bench_cfg3.json'))
print('cfg3', round(r['value']/1e6,1), 'Mframes/s', round(r['ms_per_step'],3), 'ms/step dp_ms', round(r['roofline']['kernel_ms'],3))
print('fit', r['fit_stats']['ms'], r['fit_stats']['roofline']['frac'])
print('e2e', {k: (round(v['ms'],2) if isinstance(v, dict) else '') for k, v in r.get('predict_end_to_end').items()})
PY
