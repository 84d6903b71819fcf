# round 4 (diagnostic, wrong results by construction): what the chain wave's time is made of -- ablation builds with block stamps
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( for c in 11; do for lib in profspec pa1 pa256 pa512 pa1792; do SMM_ONLY_BAND=1 timeout -k 10 120 python -c "
import sys; sys.path.insert(0,'scripts'); sys.path.insert(0,'.')
import os; os.environ['SMM_BAND']='1'
import prof_band
print('== $lib'); prof_band.run(64, 4096, $c, 1024, 'libsmmdp_$lib.so')
"; done; done ) 2>&1 | grep -v "amdgpu.ids\|wave  *[89] \|wave 1[0-5]" > gpurun_out/r4c_ablate.txt
cat gpurun_out/r4c_ablate.txt
