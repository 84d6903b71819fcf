# round 4 (diagnostic): phases of the DP kernel (prologue | forward | back-trace) of workgroup 0, profile builds
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( for c in 23; do for lib in $1; do SMM_ONLY_BAND=1 timeout -k 10 200 python -c "
import sys; sys.path.insert(0,'scripts'); sys.path.insert(0,'.')
import os; os.environ['SMM_BAND']='1'
import prof_band
print('== $lib'); prof_band.run(64, 4096, $c, 1024, 'libsmmdp_$lib.so')
"; done; done
  timeout -k 10 300 python scripts/prof_cfg3.py $1 ) 2>&1 | grep -v "amdgpu.ids\|wave  *[89] \|wave 1[0-5]" > gpurun_out/r4m.txt
cat gpurun_out/r4m.txt
