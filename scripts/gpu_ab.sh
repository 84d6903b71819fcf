# same-box A/B of two library builds (action-segmentation_amd/libsmmdp_clold.so / libsmmdp_clnew.so) on the bench corpora
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in clold clnew; do
  export SMM_LIB_PATH=$PWD/action-segmentation_amd/libsmmdp_$v.so
  unset SMM_PAIRS SMM_TRIPLES
  for s in 1000 1001 1003 1006; do timeout -k 10 300 python bench.py --workload cfg3 --seed $s --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$v seed $s', round(r['value']/1e6,1), round(r['ms_per_step'],3), round(r['roofline']['kernel_ms'],3))"; done
done; done
