# round 4: variants by name: plain timings (time_variants) and block stamps at 11 / 23 states
# usage: gpurun -- 'bash scripts/gpu_r4_d.sh "plain1 plain2 ..." "prof1 prof2 ..."'
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout -k 10 300 python scripts/time_variants.py $1
  for c in 23 11; do for lib in $2; do SMM_ONLY_BAND=1 timeout -k 10 200 python -c "
import sys; sys.path.insert(0,'scripts'); sys.path.insert(0,'.')
import os; os.environ['SMM_BAND']='1'
import prof_band
print('== $lib'); prof_band.run(64, 4096, $c, 1024, 'libsmmdp_$lib.so')
"; done; done ) 2>&1 | grep -v "amdgpu.ids\|wave  *[89] \|wave 1[0-5]" > gpurun_out/r4d.txt
cat gpurun_out/r4d.txt
