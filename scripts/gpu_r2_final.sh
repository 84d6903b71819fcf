# round 2 final pass: GPU suite, bench lines (cfg3 default, cfg1, cfg2, cfg4, cfg5 strong leg on one rank, a 2-rank rehearsal),
# rocprofv3 kernel stats and PMC passes of the default bench
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()"
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r2_final_pytest.log 2>&1 ; echo "all tests rc=$?"
tail -3 gpurun_out/r2_final_pytest.log
timeout -k 10 600 python bench.py --steps 5 --warmup 2 2>&1 | tail -1 > gpurun_out/r2_final_cfg3.json
timeout -k 10 600 python bench.py --workload cfg2 --steps 5 --warmup 2 2>&1 | tail -1 > gpurun_out/r2_final_cfg2.json
timeout -k 10 600 python bench.py --workload cfg1 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r2_final_cfg1.json
timeout -k 10 600 python bench.py --workload cfg4 --steps 5 --warmup 2 2>&1 | tail -1 > gpurun_out/r2_final_cfg4.json
timeout -k 10 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-predict-e2e --strong-leg 2>&1 | tail -1 > gpurun_out/r2_final_cfg5_strong_1rank.json
timeout -k 10 900 python bench.py --gpus 2 --backend gloo --share-gpus --steps 3 --warmup 1 --no-cpu-baseline --strong-workload cfg3 2>&1 | tail -1 > gpurun_out/r2_final_2ranks_gloo_rehearsal.json
rm -rf gpurun_out/prof_cfg3 gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg3 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-predict-e2e > gpurun_out/prof_cfg3.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-predict-e2e > gpurun_out/pmc_$c.log 2>&1
done
python scripts/pmc_summary.py gpurun_out cfg3 "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of \`python bench.py --steps 2 --warmup 1\` (scripts/gpu_r2_final.sh), kernels as of commit ${SMM_COMMIT:-unknown}" > /dev/null
rm -rf gpurun_out/prof_cfg4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg4 -- python bench.py --workload cfg4 --steps 3 --warmup 1 --no-cpu-baseline --no-predict-e2e > gpurun_out/prof_cfg4.log 2>&1
python - <<'PY'
import json, csv, glob
for w in ('cfg3', 'cfg2', 'cfg1', 'cfg4'):
    r = json.load(open('gpurun_out/r2_final_%s.json' % w))
    print(w, round(r['value']/1e6, 1), 'Mframes/s', round(r['ms_per_step'], 3), 'ms/step dp_ms', round(r['roofline']['kernel_ms'], 3),
          'frac', round(r['roofline']['frac'], 4), 'mof', round(r['mof'], 4), 'fit', round(r['fit_stats']['ms'], 3),
          r.get('cpu_baseline', {}).get('value'), r.get('cpu_factored', {}).get('value'))
r = json.load(open('gpurun_out/r2_final_cfg3.json')); print('e2e', r.get('predict_end_to_end'))
r = json.load(open('gpurun_out/r2_final_cfg4.json')); print('cfg4 logz', json.dumps(r.get('logz_fwd_bwd'))[:900])
r = json.load(open('gpurun_out/r2_final_cfg5_strong_1rank.json')); print('cfg5 strong 1 rank', json.dumps(r.get('strong_scaling'))[:700])
r = json.load(open('gpurun_out/r2_final_2ranks_gloo_rehearsal.json')); print('2 ranks (gloo, shared GPU)', r['n_gpus'], r['value'], json.dumps(r.get('strong_scaling'))[:500])
for w in ('cfg3', 'cfg4'):
    f = glob.glob('gpurun_out/prof_%s/**/*kernel_stats.csv' % w, recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r['TotalDurationNs']))
    for r in rows:
        if 'smm_' in r['Name']:
            print(w, r['Name'][:70], '| calls', r['Calls'], '| avg_us', round(float(r['AverageNs'])/1e3, 1))
PY
