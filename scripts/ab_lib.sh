# same-box A/B of two builds of libsmmdp.so (SMM_LIB_PATH) through bench.py: bash scripts/ab_lib.sh [workloads...]
cd $GRAFT_REPO_ROOT
for w in ${@:-cfg2 cfg3}; do
  for rep in 1 2; do
    for lib in libsmmdp_prev.so libsmmdp.so; do
      SMM_LIB_PATH=$GRAFT_REPO_ROOT/action-segmentation_amd/$lib timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-predict-e2e --no-strong-leg --second-seed -1 2>/dev/null | tail -1 | python -c "
import sys, json; j=json.loads(sys.stdin.read()); print('$w $lib', round(j['value']/1e6,1), 'M', round(j['ms_per_step'],3), 'ms/step dp', round(j['roofline']['kernel_ms'],4))"
    done
  done
done
