# round 2, first GPU pass: new multi-rank / cli tests, full GPU suite, bench at the new default seed, kernel stats
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r2a_pytest_new.log 2>&1 ; echo "new tests rc=$?" 
tail -5 gpurun_out/r2a_pytest_new.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2a_pytest_all.log 2>&1 ; echo "all tests rc=$?"
tail -3 gpurun_out/r2a_pytest_all.log
timeout -k 10 600 python bench.py --steps 5 --warmup 2 2> gpurun_out/r2a_bench_cfg3.err | tail -1 > gpurun_out/r2a_bench_cfg3.json
python - <<'PY'
import json
r=json.load(open('gpurun_out/r2a_bench_cfg3.json'))
print('cfg3', round(r['value']/1e6,1), 'Mframes/s', round(r['ms_per_step'],3), 'ms/step dp_ms', round(r['roofline']['kernel_ms'],3), r['config']['workload'])
print('e2e', r.get('predict_end_to_end'))
print('cpu', r.get('cpu_baseline',{}).get('value'), r.get('cpu_factored'))
PY
