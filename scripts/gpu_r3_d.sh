set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_fullsize.py -q -x -k "band or whole_corpus or longest" > gpurun_out/r3d_pytest.log 2>&1 ; echo "tests rc=$?"
tail -8 gpurun_out/r3d_pytest.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-predict-e2e 2>gpurun_out/r3d_cfg3.err | tail -1 > gpurun_out/r3d_cfg3.json
timeout -k 10 300 python bench.py --workload cfg1 --steps 10 --warmup 3 --no-cpu-baseline --no-predict-e2e 2>gpurun_out/r3d_cfg1.err | tail -1 > gpurun_out/r3d_cfg1.json
timeout -k 10 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-predict-e2e --strong-leg 2>gpurun_out/r3d_cfg5.err | tail -1 > gpurun_out/r3d_cfg5.json
python - <<'PY'
import json
for w in ('cfg3', 'cfg1', 'cfg5'):
    try:
        r = json.load(open('gpurun_out/r3d_%s.json' % w))
        print(w, round(r['value']/1e6, 1), 'Mframes/s', round(r['ms_per_step'], 3), 'ms dp', round(r['roofline']['kernel_ms'], 3), 'mof', r['mof'], 'other', r.get('other_draw', {}).get('dp_kernel_ms'))
        if 'strong_scaling' in r:
            ss = r['strong_scaling']; print('   strong', round(ss['value']/1e6,1), ss['ms_per_step'], ss['dp_kernel_ms_max_over_ranks'])
    except Exception as e:
        print(w, 'failed', e)
PY
