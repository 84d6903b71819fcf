set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 200 python scripts/probe_predict_cfg4.py > gpurun_out/r3b_probe_cfg4.txt 2>&1 ; echo "probe rc=$?"
tail -30 gpurun_out/r3b_probe_cfg4.txt
# band mode: parity first
timeout -k 10 600 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_random_sweeps.py -q -x > gpurun_out/r3b_pytest_vit.log 2>&1 ; echo "viterbi tests rc=$?"
tail -15 gpurun_out/r3b_pytest_vit.log
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -q -x --durations=12 -k "not log_partition" > gpurun_out/r3b_pytest_full.log 2>&1 ; echo "fullsize rc=$?"
tail -25 gpurun_out/r3b_pytest_full.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-predict-e2e 2>gpurun_out/r3b_cfg3.err | tail -1 > gpurun_out/r3b_cfg3_band.json
SMM_BAND=0 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-predict-e2e 2>gpurun_out/r3b_cfg3_old.err | tail -1 > gpurun_out/r3b_cfg3_old.json
timeout -k 10 300 python bench.py --workload cfg1 --steps 5 --warmup 2 --no-cpu-baseline --no-predict-e2e 2>gpurun_out/r3b_cfg1.err | tail -1 > gpurun_out/r3b_cfg1_band.json
python - <<'PY'
import json
for w in ('cfg3_band', 'cfg3_old', 'cfg1_band'):
    try:
        r = json.load(open('gpurun_out/r3b_%s.json' % w))
        print(w, round(r['value']/1e6, 1), 'Mframes/s', round(r['ms_per_step'], 3), 'ms dp', round(r['roofline']['kernel_ms'], 3), 'mof', r['mof'], 'other', r.get('other_draw', {}).get('dp_kernel_ms'))
    except Exception as e:
        print(w, 'failed', e)
PY
