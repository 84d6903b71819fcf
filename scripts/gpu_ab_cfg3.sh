# development aid: bench.py's cfg3 step with full-library variants / environment settings on ONE box
# usage: gpurun -- 'bash scripts/gpu_ab_cfg3.sh "tag[:ENV=val]" ...'
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/ab_cfg3.txt
for rep in 1 2; do for spec in "$@"; do
  tag=${spec%%:*}; envs=""; [ "$spec" != "$tag" ] && envs=${spec#*:}
  env ${envs//,/ } SMM_LIB_PATH=$PWD/action-segmentation_amd/libsmmdp_$tag.so timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-predict-e2e --second-seed -1 2>/dev/null | grep '^{"metric' | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('$spec', round(r['value']/1e6,1), 'Mframes/s', round(r['ms_per_step'],3), 'ms/step; DP launches/step', ro['launches_per_step'], 'mean', round(ro['kernel_ms'],3), 'longest', round(ro['kernel_ms_longest_launch'],3))" >> gpurun_out/ab_cfg3.txt
done; done
cat gpurun_out/ab_cfg3.txt
