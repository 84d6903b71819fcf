# round 5 acceptance pass: smoke, GPU suite, bench lines (cfg3 default with its strong leg, cfg2, cfg1, cfg4, refdef, cfg5 strong leg
# over a one-rank RCCL group, 2- and 4-rank gloo rehearsals), rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes of the default
# bench, the FETCH_SIZE calibration runs, SQ counters, the soak
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${SMM_TAG:-r5_final}
export SMM_TAG=$T
python -c "import __graft_entry__ as g; g.smoke()"
timeout -k 10 1000 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/${T}_pytest.log 2>&1 ; echo "all tests rc=$?"
tail -14 gpurun_out/${T}_pytest.log
( time timeout -k 10 600 python bench.py --steps 20 --warmup 5 2>gpurun_out/${T}_cfg3.err | tail -1 > gpurun_out/${T}_cfg3.json ) 2>&1 | tail -3
timeout -k 10 600 python bench.py --workload cfg2 --steps 20 --warmup 5 --no-strong-leg 2>gpurun_out/${T}_cfg2.err | tail -1 > gpurun_out/${T}_cfg2.json
timeout -k 10 600 python bench.py --workload cfg1 --steps 20 --warmup 5 --no-strong-leg 2>gpurun_out/${T}_cfg1.err | tail -1 > gpurun_out/${T}_cfg1.json
timeout -k 10 600 python bench.py --workload cfg4 --steps 20 --warmup 5 --no-strong-leg 2>gpurun_out/${T}_cfg4.err | tail -1 > gpurun_out/${T}_cfg4.json
timeout -k 10 600 python bench.py --workload refdef --steps 20 --warmup 5 --no-predict-e2e --no-strong-leg 2>gpurun_out/${T}_refdef.err | tail -1 > gpurun_out/${T}_refdef.json
SMM_DIST_SINGLE_RANK=1 timeout -k 10 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-predict-e2e --scaling strong 2>gpurun_out/${T}_cfg5.err | tail -1 > gpurun_out/${T}_cfg5_strong_1rank_rccl.json
timeout -k 10 900 python bench.py --gpus 2 --backend gloo --share-gpus --steps 3 --warmup 1 --no-cpu-baseline --strong-workload cfg3 2>gpurun_out/${T}_2r.err | tail -1 > gpurun_out/${T}_2ranks_gloo_rehearsal.json
timeout -k 10 900 python bench.py --gpus 4 --backend gloo --share-gpus --steps 3 --warmup 1 --no-cpu-baseline --strong-workload cfg3 2>gpurun_out/${T}_4r.err | tail -1 > gpurun_out/${T}_4ranks_gloo_rehearsal.json
rm -rf gpurun_out/prof_cfg3 gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_sq gpurun_out/pmc_calib_ubench gpurun_out/pmc_calib_emission
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg3 -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-predict-e2e --no-strong-leg --second-seed -1 > gpurun_out/prof_cfg3.log 2>&1
grep "^{\"metric" gpurun_out/prof_cfg3.log | tail -1 > gpurun_out/${T}_cfg3_under_rocprof.json
# FETCH_SIZE calibration (known byte counts in the library's own access shapes), then the passes of the bench itself
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_calib_ubench -- scripts/ubench/_bin/fetch_calib > gpurun_out/pmc_calib_ubench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_calib_emission -- python scripts/probe_emission_alone.py > gpurun_out/pmc_calib_emission.log 2>&1
python scripts/pmc_calibrate.py gpurun_out > gpurun_out/${T}_pmc_calibration.txt 2>&1; cat gpurun_out/${T}_pmc_calibration.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-predict-e2e --no-strong-leg --second-seed -1 > gpurun_out/pmc_$c.log 2>&1
done
python scripts/pmc_summary.py gpurun_out cfg3 "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of \`python bench.py --steps 2 --warmup 1\` (scripts/gpu_r5_final.sh), FETCH_SIZE scaled by the calibrated per-shape factors of profiles/pmc_calibration.json; kernels as of commit ${SMM_COMMIT:-unknown}" > /dev/null
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
timeout -k 10 240 python scripts/soak_band.py 100 5 > gpurun_out/${T}_soak_band.txt 2>&1; tail -1 gpurun_out/${T}_soak_band.txt
timeout -k 10 200 python scripts/soak_band.py 60 6 2 512 32 >> gpurun_out/${T}_soak_band.txt 2>&1; tail -1 gpurun_out/${T}_soak_band.txt
python - <<'PY'
import json
import os
T = os.environ.get('SMM_TAG', 'r5_final')
for w in ('cfg3', 'cfg2', 'cfg1', 'cfg4', 'refdef'):
    try:
        r = json.load(open('gpurun_out/%s_%s.json' % (T, w)))
        print(w, round(r['value']/1e6, 1), 'Mframes/s', round(r['ms_per_step'], 3), 'ms/step', r['scaling'], 'dp_ms', round(r['roofline']['kernel_ms'], 3),
              'crit', r['roofline'].get('critical_launch_ms'), 'frac', round(r['roofline']['frac'], 4), 'mof', round(r['mof'], 4),
              'cpu', r.get('cpu_baseline', {}).get('value'), r.get('cpu_factored', {}).get('value'), 'split', r.get('time_split'),
              'parity', {k: v for k, v in r.get('parity', {}).items() if k not in ('what', 'grad_tolerance')})
    except Exception as e:
        print(w, 'failed', e)
r = json.load(open('gpurun_out/%s_cfg3.json' % T))
for k in ('predict_end_to_end', 'host_features', 'reference_default', 'evaluation', 'fit_stats'):
    v = r.get(k)
    if isinstance(v, dict): v = {a: b for a, b in v.items() if a not in ('what', 'stats', 'roofline')}
    print(k, v)
s = r.get('strong_scaling') or {}
print('strong leg in the default line:', s.get('value'), s.get('ms_per_step'), s.get('dp_kernel_ms_max_over_ranks'))
for n in ('cfg5_strong_1rank_rccl', '2ranks_gloo_rehearsal', '4ranks_gloo_rehearsal'):
    try:
        r = json.load(open('gpurun_out/%s_%s.json' % (T, n)))
        print(n, r['scaling'], round(r['value']/1e6, 1), 'Mframes/s', round(r['ms_per_step'], 3), 'ms/step', r.get('backend'), '|', r['config']['workload'][:80])
    except Exception as e:
        print(n, 'failed', e)
PY
