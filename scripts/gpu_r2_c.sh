# round 2: whole GPU suite + cfg3 bench
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r2c_pytest_all.log 2>&1 ; echo "all tests rc=$?"
tail -12 gpurun_out/r2c_pytest_all.log
timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/r2c_bench_cfg3.err | tail -1 > gpurun_out/r2c_bench_cfg3.json
python - <<'PY'
import json
r=json.load(open('gpurun_out/r2c_bench_cfg3.json'))
print('cfg3', round(r['value']/1e6,1), 'Mframes/s', round(r['ms_per_step'],3), 'ms/step dp_ms', round(r['roofline']['kernel_ms'],3))
print('e2e', r.get('predict_end_to_end'))
PY
