# development aid: cfg4's training-step kernels (log-partition forward / backward) with full-library variants on ONE box
# usage: gpurun -- 'bash scripts/gpu_ab_logz.sh lz0 lzbar ...'
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/ab_logz.txt
for rep in 1 2; do for tag in "$@"; do
  SMM_LIB_PATH=$PWD/action-segmentation_amd/libsmmdp_$tag.so timeout -k 10 300 python bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu-baseline --no-predict-e2e 2>/dev/null | grep '^{"metric' | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); z=r['logz_fwd_bwd']; print('$tag', 'packed', round(z['packed']['ms'],3), 'ms  per_batch', round(z['per_batch']['ms'],2), 'ms  kernels: logz fwd(+bwd launch)', round(z['kernels']['logz_fwd_ms'],4), 'marginals etc', round(z['kernels']['logz_bwd_ms'],4), ' decode step', round(r['ms_per_step'],4))" >> gpurun_out/ab_logz.txt
done; done
cat gpurun_out/ab_logz.txt
