set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py --steps 5 --warmup 2 2>&1 | tail -1 > gpurun_out/bench_cfg3.json
python bench.py --workload cfg2 --steps 5 --warmup 2 2>&1 | tail -1 > gpurun_out/bench_cfg2.json
python bench.py --workload cfg1 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_cfg1.json
python bench.py --workload cfg4 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_cfg4.json
rm -rf gpurun_out/prof_cfg3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg3 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_cfg3.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$c.log 2>&1
done
python scripts/pmc_summary.py gpurun_out cfg3 > /dev/null
for w in cfg1 cfg2 cfg3 cfg4; do python -c "
import json; r=json.load(open('gpurun_out/bench_$w.json')); print('$w', round(r['value']/1e6,1), 'Mframes/s', round(r['ms_per_step'],3), 'ms/step dp_ms', round(r['roofline']['kernel_ms'],3), 'frac', round(r['roofline']['frac'],4), 'mof', round(r['mof'],4), r.get('cpu_baseline',{}).get('value'), r.get('cpu_factored',{}).get('value'))"; done
