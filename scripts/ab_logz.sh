# same-box A/B of two builds of libsmmdp.so on the cfg4 training step (bench.py --workload cfg4: logz_fwd_bwd)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in libsmmdp_prev.so libsmmdp.so; do
    SMM_LIB_PATH=$GRAFT_REPO_ROOT/action-segmentation_amd/$lib timeout -k 10 300 python bench.py --workload cfg4 --steps 5 --warmup 2 --no-cpu-baseline --no-predict-e2e --no-strong-leg 2>/dev/null | tail -1 | python -c "
import sys, json; j=json.loads(sys.stdin.read()); l=j['logz_fwd_bwd']; print('$lib packed', l['packed']['passes_ms'], 'per-batch ms', round(l['ms_per_batch'],3), 'kernels fwd', round(l['kernels']['logz_fwd_ms'],3), 'bwd', round(l['kernels']['logz_bwd_ms'],3), 'decode step', round(j['ms_per_step'],3))"
  done
done
