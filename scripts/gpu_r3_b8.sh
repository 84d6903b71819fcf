# B = 8 BAND kernel: whole GPU suite, then bench lines + the cfg3 stamps
set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -x -m gpu > gpurun_out/b8_tests.txt 2>&1; tail -3 gpurun_out/b8_tests.txt
timeout -k 10 300 python scripts/prof_cfg3.py base profbase > gpurun_out/b8_prof.txt 2>&1
for w in cfg3 cfg1 cfg2; do timeout -k 10 400 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>gpurun_out/b8_$w.err | tail -1 > gpurun_out/b8_$w.json; done
SMM_DIST_SINGLE_RANK=1 timeout -k 10 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-predict-e2e --strong-leg --scaling strong 2>gpurun_out/b8_cfg5.err | tail -1 > gpurun_out/b8_cfg5.json
python - <<'PY'
import json
for w in ('cfg3','cfg1','cfg2','cfg5'):
    try:
        r=json.load(open('gpurun_out/b8_%s.json'%w)); print(w, round(r['value']/1e6,1),'Mframes/s', round(r['ms_per_step'],3),'ms', 'dp', round(r['roofline']['kernel_ms'],3) if 'roofline' in r else None, r.get('parity'))
    except Exception as e: print(w,'failed',e)
PY
