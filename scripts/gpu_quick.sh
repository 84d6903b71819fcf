# quick correctness + timing loop for kernel work: parity tests, then the bench lines without the CPU baseline
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_module.py tests/test_gpu_model.py -x -q > gpurun_out/pytest_quick.log 2>&1; tail -5 gpurun_out/pytest_quick.log
for w in cfg3 cfg2 cfg1; do
  timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/q_$w.json
  python -c "
import json; r=json.load(open('gpurun_out/q_$w.json')); print('$w', round(r['value']/1e6,1), 'Mframes/s', round(r['ms_per_step'],3), 'ms/step dp_ms', round(r['roofline']['kernel_ms'],3), 'frac', round(r['roofline']['frac'],4), 'mof', round(r['mof'],4))"
done
rm -rf gpurun_out/prof_q
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_q.log 2>&1
grep -h "smm_" gpurun_out/prof_q/*/*kernel_stats.csv | cut -c1-60,150-260
