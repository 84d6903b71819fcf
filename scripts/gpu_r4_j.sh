# round 4 (diagnostic, ablation builds: wrong results by construction): what a hand-over block's time is made of
# usage: gpurun -- 'bash scripts/gpu_r4_j.sh "<tags to time>"'
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 500 python scripts/time_variants.py $1 2>&1 | grep -v "amdgpu.ids\|pass 0" > gpurun_out/r4j.txt
cat gpurun_out/r4j.txt
