# round 4: source dominance in band 0 (DOM): parity tests of the K > 512 kernels through the shipped library, then timings
# usage: gpurun -- 'bash scripts/gpu_r4_i.sh "<tags to time>" "<prof tags>"'
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_viterbi.py -m gpu -q -x > gpurun_out/r4i_tests.txt 2>&1; rc=$?
tail -4 gpurun_out/r4i_tests.txt; echo "tests rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
( timeout -k 10 300 python scripts/time_variants.py $1
  for c in 23 11; do for lib in $2; do SMM_ONLY_BAND=1 timeout -k 10 200 python -c "
import sys; sys.path.insert(0,'scripts'); sys.path.insert(0,'.')
import os; os.environ['SMM_BAND']='1'
import prof_band
print('== $lib'); prof_band.run(64, 4096, $c, 1024, 'libsmmdp_$lib.so')
"; done; done
  timeout -k 10 300 python scripts/prof_cfg3.py $1 ) 2>&1 | grep -v "amdgpu.ids\|wave  *[89] \|wave 1[0-5]\|pass 0" > gpurun_out/r4i.txt
cat gpurun_out/r4i.txt
