# round 4: block stamps of the shipped BAND kernel (diagnostic -DSMM_PROFILE builds of the same source: scripts/build_variants.sh
# fin "" pfin1 "-DSMM_PROFILE=1" pfin2 "-DSMM_PROFILE=2"): CrossTask-like lattices (64 x 4096 frames, K = 1024) and the cfg3 corpus
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout -k 10 300 python scripts/time_variants.py fin
  for c in 23 11; do SMM_ONLY_BAND=1 timeout -k 10 200 python -c "
import sys; sys.path.insert(0,'scripts'); sys.path.insert(0,'.')
import os; os.environ['SMM_BAND']='1'
import prof_band
print('== pfin1 (-DSMM_PROFILE=1)'); prof_band.run(64, 4096, $c, 1024, 'libsmmdp_pfin1.so')
"; done
  timeout -k 10 300 python scripts/prof_cfg3.py fin pfin1
  SMM_PROF_LAST=1 timeout -k 10 300 python scripts/prof_cfg3.py pfin2 ) 2>&1 | grep -v "amdgpu.ids\|wave  *[89] \|wave 1[0-5]\|pass 0" > gpurun_out/r4_stamps.txt
cat gpurun_out/r4_stamps.txt
