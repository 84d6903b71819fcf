# round 4: the shipped library through the Viterbi GPU tests, then BAND-only variants: timings, who arrives last (-DSMM_PROFILE=2)
# usage: gpurun -- 'bash scripts/gpu_r4_o.sh "<tags to time>" "<PROFILE=2 tags>"'
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_viterbi.py -m gpu -q -x > gpurun_out/r4o_tests.txt 2>&1; rc=$?
tail -3 gpurun_out/r4o_tests.txt; echo "tests rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
( timeout -k 10 300 python scripts/time_variants.py $1
  timeout -k 10 300 python scripts/prof_cfg3.py $1
  SMM_PROF_LAST=1 timeout -k 10 300 python scripts/prof_cfg3.py $2 ) 2>&1 | grep -v "amdgpu.ids\|wave  *[89] \|wave 1[0-5]\|pass 0" > gpurun_out/r4o.txt
cat gpurun_out/r4o.txt
