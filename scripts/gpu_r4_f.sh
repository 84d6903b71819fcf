# round 4: whole bench steps, round 3's library (libsmmdp_prev.so) against this round's, same box, by workload; then the split sweep
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash scripts/ab_lib.sh cfg3 cfg2 cfg1 cfg4 refdef 2>&1 | grep -v amdgpu.ids > gpurun_out/r4f_ab.txt
cat gpurun_out/r4f_ab.txt
timeout -k 10 500 python scripts/sweep_split.py 7 2>&1 | grep -v amdgpu.ids > gpurun_out/r4f_split_sweep.txt
cat gpurun_out/r4f_split_sweep.txt
