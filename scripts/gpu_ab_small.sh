# development aid: the small-ring workloads (cfg2, cfg4, refdef) with full-library variants on ONE box
# usage: gpurun -- 'bash scripts/gpu_ab_small.sh full b8r4 ...'
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/ab_small.txt
for rep in 1 2; do for tag in "$@"; do for w in cfg2 cfg4 refdef; do
  SMM_LIB_PATH=$PWD/action-segmentation_amd/libsmmdp_$tag.so timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-predict-e2e 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$tag', '$w', round(r['value']/1e6,1), 'Mframes/s', round(r['ms_per_step'],4), 'ms  dp', round(r['roofline']['kernel_ms'],4))" >> gpurun_out/ab_small.txt
done; done; done
cat gpurun_out/ab_small.txt
