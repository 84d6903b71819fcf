# round 4: the Viterbi GPU tests through the shipped library (all ring sizes), then BAND-only variants timed, then cfg2 / cfg1 bench lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_fullsize.py tests/test_gpu_random_sweeps.py -m gpu -q -x > gpurun_out/r4z_tests.txt 2>&1; rc=$?
tail -3 gpurun_out/r4z_tests.txt; echo "tests rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
( timeout -k 10 300 python scripts/time_variants.py $1
  timeout -k 10 300 python scripts/prof_cfg3.py $1 ) 2>&1 | grep -v "amdgpu.ids\|pass 0" > gpurun_out/r4z.txt
cat gpurun_out/r4z.txt
for w in cfg2 cfg1; do timeout -k 10 200 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-predict-e2e 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$w %.3f ms/step, DP %.3f ms' % (r['ms_per_step'], r['roofline']['kernel_ms']))"; done
