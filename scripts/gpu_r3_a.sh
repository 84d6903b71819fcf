# round 3, first GPU pass: the GPU suite with the new parity / RCCL tests, the cfg4 predict() probe, bench lines with the parity object
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/r3a_pytest.log 2>&1 ; echo "tests rc=$?"
tail -15 gpurun_out/r3a_pytest.log
timeout -k 10 300 python scripts/probe_predict_cfg4.py > gpurun_out/r3a_probe_cfg4.txt 2>&1 ; echo "probe rc=$?"
timeout -k 10 600 python bench.py --steps 5 --warmup 2 2>gpurun_out/r3a_cfg3.err | tail -1 > gpurun_out/r3a_cfg3.json
timeout -k 10 600 python bench.py --workload cfg4 --steps 5 --warmup 2 2>gpurun_out/r3a_cfg4.err | tail -1 > gpurun_out/r3a_cfg4.json
python - <<'PY'
import json
for w in ('cfg3', 'cfg4'):
    try:
        r = json.load(open('gpurun_out/r3a_%s.json' % w))
        print(w, round(r['value']/1e6, 1), 'Mframes/s', round(r['ms_per_step'], 3), 'ms dp', round(r['roofline']['kernel_ms'], 3), 'parity', r.get('parity'))
        print('   e2e', r.get('predict_end_to_end'))
    except Exception as e:
        print(w, 'failed', e)
PY
