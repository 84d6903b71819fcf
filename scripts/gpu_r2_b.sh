# round 2: log-partition kernel v2 -- parity, timing probe, cfg4 bench
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_module.py tests/test_gpu_model.py tests/test_gpu_random_sweeps.py -x -q -m gpu -k "logz or likelihood or partition or fit or packed or edge or sweep" > gpurun_out/r2b_pytest.log 2>&1 ; echo "tests rc=$?"
tail -25 gpurun_out/r2b_pytest.log
timeout -k 10 300 python scripts/perf_probe_logz.py > gpurun_out/r2b_probe_logz.txt 2>&1; cat gpurun_out/r2b_probe_logz.txt
timeout -k 10 600 python bench.py --workload cfg4 --steps 5 --warmup 2 2> gpurun_out/r2b_bench_cfg4.err | tail -1 > gpurun_out/r2b_bench_cfg4.json
python - <<'PY'
import json
r=json.load(open('gpurun_out/r2b_bench_cfg4.json'))
print('cfg4 decode', round(r['value']/1e6,1), 'Mframes/s')
print(json.dumps(r.get('logz_fwd_bwd'), indent=1))
PY
