# round 4 (diagnostic): segment stamps of the BAND pushers' block (-DSMM_PROFILE=3)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( for c in 23 11; do for lib in $1; do SMM_ONLY_BAND=1 SMM_PROF_SEG=1 timeout -k 10 200 python -c "
import sys; sys.path.insert(0,'scripts'); sys.path.insert(0,'.')
import os; os.environ['SMM_BAND']='1'
import prof_band
print('== $lib'); prof_band.run(64, 4096, $c, 1024, 'libsmmdp_$lib.so')
"; done; done ) 2>&1 | grep -v "amdgpu.ids\|wave  *[89] \|wave 1[0-5]" > gpurun_out/r4k.txt
cat gpurun_out/r4k.txt
