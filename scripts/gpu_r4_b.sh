# round 4: the speculative-transition kernel by state count: plain timings (r3 / spec) and block stamps
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout -k 10 300 python scripts/time_variants.py r3 spec
  for c in 23 11; do for lib in profspec; do SMM_ONLY_BAND=1 timeout -k 10 200 python -c "
import sys; sys.path.insert(0,'scripts'); sys.path.insert(0,'.')
import os; os.environ['SMM_BAND']='1'
import prof_band
print('== $lib'); prof_band.run(64, 4096, $c, 1024, 'libsmmdp_$lib.so')
"; done; done
  timeout -k 10 300 python scripts/prof_cfg3.py r3 spec profspec ) 2>&1 | grep -v "amdgpu.ids\|wave  *[89] \|wave 1[0-5]" > gpurun_out/r4b_stamps.txt
cat gpurun_out/r4b_stamps.txt
