set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_graph.py tests/test_gpu_fullsize.py tests/test_gpu_model.py -q -x -k "not log_partition" > gpurun_out/r3s_pytest.log 2>&1 ; echo "tests rc=$?"
tail -8 gpurun_out/r3s_pytest.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-predict-e2e 2>gpurun_out/r3s_cfg3.err | tail -1 > gpurun_out/r3s_cfg3.json
SMM_NO_SPLIT=1 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-predict-e2e 2>gpurun_out/r3s_cfg3n.err | tail -1 > gpurun_out/r3s_cfg3_nosplit.json
SMM_VERBOSE=1 timeout -k 10 300 python bench.py --workload cfg2 --steps 20 --warmup 3 --no-cpu-baseline --no-predict-e2e 2>gpurun_out/r3s_cfg2.err | tail -1 > gpurun_out/r3s_cfg2.json
python - <<'PY'
import json
for w in ('cfg3', 'cfg3_nosplit', 'cfg2'):
    try:
        r = json.load(open('gpurun_out/r3s_%s.json' % w))
        print(w, round(r['value']/1e6, 1), 'Mframes/s', round(r['ms_per_step'], 3), 'ms dp', round(r['roofline']['kernel_ms'], 3), 'mof', r['mof'], 'other', r.get('other_draw', {}).get('ms_per_step'))
    except Exception as e:
        print(w, 'failed', e)
PY
tail -3 gpurun_out/r3s_cfg3.err
