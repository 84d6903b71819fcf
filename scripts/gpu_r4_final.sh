# round 4 acceptance pass: smoke, GPU suite, bench lines (cfg3 default, cfg2, cfg1, cfg4, refdef, cfg5 strong leg on one rank over a
# one-rank RCCL group, a 2-rank gloo rehearsal), rocprofv3 kernel stats + PMC passes of the default bench, SQ counters, split sweep
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${SMM_TAG:-r4_final}
export SMM_TAG=$T
python -c "import __graft_entry__ as g; g.smoke()"
timeout -k 10 1000 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/${T}_pytest.log 2>&1 ; echo "all tests rc=$?"
tail -14 gpurun_out/${T}_pytest.log
timeout -k 10 600 python bench.py --steps 20 --warmup 3 2>gpurun_out/${T}_cfg3.err | tail -1 > gpurun_out/${T}_cfg3.json
timeout -k 10 600 python bench.py --workload cfg2 --steps 20 --warmup 3 2>gpurun_out/${T}_cfg2.err | tail -1 > gpurun_out/${T}_cfg2.json
timeout -k 10 600 python bench.py --workload cfg1 --steps 20 --warmup 3 2>gpurun_out/${T}_cfg1.err | tail -1 > gpurun_out/${T}_cfg1.json
timeout -k 10 600 python bench.py --workload cfg4 --steps 20 --warmup 3 2>gpurun_out/${T}_cfg4.err | tail -1 > gpurun_out/${T}_cfg4.json
timeout -k 10 600 python bench.py --workload refdef --steps 20 --warmup 3 --no-predict-e2e 2>gpurun_out/${T}_refdef.err | tail -1 > gpurun_out/${T}_refdef.json
SMM_DIST_SINGLE_RANK=1 timeout -k 10 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-predict-e2e --strong-leg --scaling strong 2>gpurun_out/${T}_cfg5.err | tail -1 > gpurun_out/${T}_cfg5_strong_1rank_rccl.json
timeout -k 10 900 python bench.py --gpus 2 --backend gloo --share-gpus --steps 3 --warmup 1 --no-cpu-baseline --strong-workload cfg3 2>gpurun_out/${T}_2r.err | tail -1 > gpurun_out/${T}_2ranks_gloo_rehearsal.json
rm -rf gpurun_out/prof_cfg3 gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_sq
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg3 -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-predict-e2e --second-seed -1 > gpurun_out/prof_cfg3.log 2>&1
grep "^{\"metric" gpurun_out/prof_cfg3.log | tail -1 > gpurun_out/${T}_cfg3_under_rocprof.json
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-predict-e2e --second-seed -1 > gpurun_out/pmc_$c.log 2>&1
done
python scripts/pmc_summary.py gpurun_out cfg3 "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of \`python bench.py --steps 2 --warmup 1\` (scripts/gpu_r4_final.sh), kernels as of commit ${SMM_COMMIT:-unknown}" > /dev/null
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc_sq -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-predict-e2e --no-strong-leg --second-seed -1 > gpurun_out/pmc_sq.log 2>&1
python - > gpurun_out/${T}_sq_counters.txt <<'PY'
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob('gpurun_out/pmc_sq/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:70]
        if 'smm_' not in k: continue
        rows[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVES': cnt[k] += 1
for k, d in rows.items():
    n = cnt[k] or 1
    print(k, 'launches', n, {c: round(v / n) for c, v in d.items()})
PY
cat gpurun_out/${T}_sq_counters.txt
timeout -k 10 600 python scripts/sweep_split.py > gpurun_out/${T}_split_sweep.txt 2>&1; echo "sweep rc=$?"
timeout -k 10 200 python scripts/probe_viterbi_call.py > gpurun_out/${T}_viterbi_call.txt 2>&1
timeout -k 10 200 python scripts/probe_predict_refdef.py > gpurun_out/${T}_predict_refdef.txt 2>&1
timeout -k 10 200 python scripts/probe_predict_fused.py > gpurun_out/${T}_predict_fused.txt 2>&1
python - <<'PY'
import json
import os
T = os.environ.get('SMM_TAG', 'r4_final')
for w in ('cfg3', 'cfg2', 'cfg1', 'cfg4', 'refdef'):
    try:
        r = json.load(open('gpurun_out/%s_%s.json' % (T, w)))
        print(w, round(r['value']/1e6, 1), 'Mframes/s', round(r['ms_per_step'], 3), 'ms/step dp_ms', round(r['roofline']['kernel_ms'], 3),
              'crit', r['roofline'].get('critical_launch_ms'), 'frac', round(r['roofline']['frac'], 4), 'mof', round(r['mof'], 4),
              'cpu', r.get('cpu_baseline', {}).get('value'), r.get('cpu_factored', {}).get('value'),
              'parity', {k: v for k, v in r.get('parity', {}).items() if k not in ('what', 'grad_tolerance')})
    except Exception as e:
        print(w, 'failed', e)
r = json.load(open('gpurun_out/%s_cfg3.json' % T))
for k in ('predict_end_to_end', 'host_features', 'reference_default', 'other_draw', 'evaluation', 'fit_stats'):
    v = r.get(k)
    if isinstance(v, dict): v = {a: b for a, b in v.items() if a not in ('what', 'stats', 'roofline')}
    print(k, v)
try:
    r = json.load(open('gpurun_out/%s_cfg5_strong_1rank_rccl.json' % T))
    print('cfg5 strong 1 rank', round(r['value']/1e6, 1), 'Mframes/s', round(r['ms_per_step'], 3), 'ms/step', r.get('backend'))
except Exception as e:
    print('cfg5 failed', e)
PY
