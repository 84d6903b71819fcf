# round 5, first GPU call: the suite, the default bench line and the per-state-count probe at the start of the round
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${SMM_TAG:-r5_base}
python -c "import __graft_entry__ as g; g.smoke()" || exit 1
timeout -k 10 1000 python -m pytest tests -q -m gpu -x --durations=8 > gpurun_out/${T}_pytest.log 2>&1 ; echo "all tests rc=$?"
tail -14 gpurun_out/${T}_pytest.log
( time timeout -k 10 600 python bench.py --steps 20 --warmup 5 2>gpurun_out/${T}_cfg3.err | tail -1 > gpurun_out/${T}_cfg3.json ) 2>&1 | tail -3
cp action-segmentation_amd/libsmmdp.so action-segmentation_amd/libsmmdp_base.so
timeout -k 10 300 python scripts/time_variants.py base > gpurun_out/${T}_lattices.txt 2>&1
cat gpurun_out/${T}_lattices.txt
python - <<'PY'
import json, os
T = os.environ.get('SMM_TAG', 'r5_base')
r = json.load(open('gpurun_out/%s_cfg3.json' % T))
print('cfg3', round(r['value']/1e6, 1), 'Mframes/s', round(r['ms_per_step'], 3), 'ms/step', r['scaling'], 'crit', r['roofline'].get('critical_launch_ms'),
      'parity', {k: v for k, v in r.get('parity', {}).items() if k not in ('what',)})
s = r.get('strong_scaling') or {}
print('strong', s.get('value'), s.get('ms_per_step'), s.get('dp_kernel_ms_max_over_ranks'))
for k in ('predict_end_to_end', 'reference_default'):
    v = r.get(k)
    if isinstance(v, dict): v = {a: b for a, b in v.items() if a not in ('what', 'stats', 'roofline')}
    print(k, v)
PY
