cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for lib in libsmmdp_ns0.so libsmmdp_ns0s.so; do
SMM_LIB_PATH=$PWD/action-segmentation_amd/$lib timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-predict-e2e --no-strong-leg --second-seed -1 2>/dev/null | tail -1 | python -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$lib', round(j['ms_per_step'],3), {k: (round(r.get(k),3) if r.get(k) else None) for k in ('kernel_ms','launches_per_step','kernel_ms_longest_launch','critical_launch_ms','rest_launch_ms')})"
done
