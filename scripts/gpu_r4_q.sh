# round 4 (diagnostic): a pusher's busy cycles by class of block (-DSMM_PROFILE=4) on the cfg3 corpus
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
SMM_PROF_CLS=1 timeout -k 10 300 python scripts/prof_cfg3.py $1 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r4q.txt
cat gpurun_out/r4q.txt
