# round 4, first pass: the GPU suite with the speculative-transition kernel (new parity tests included), then same-box A/B of
# round 3's BAND kernel (r3) against this round's (spec; nospec = the same source with -DSMM_SPEC=0) on the CrossTask-like
# lattices and on the bench's cfg3 corpus, with block stamps.  usage: gpurun -- 'bash scripts/gpu_r4_a.sh'
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r4a_tests.txt 2>&1; rc=$?
tail -5 gpurun_out/r4a_tests.txt; echo "tests rc=$rc"
if [ $rc -ge 124 ]; then echo "test run was killed: stopping"; exit $rc; fi
( timeout -k 10 400 python scripts/time_variants.py r3 nospec spec ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r4a_variants.txt; rc=${PIPESTATUS[0]}
cat gpurun_out/r4a_variants.txt | grep -v "cycles per block" ; echo "variants rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
( timeout -k 10 400 python scripts/prof_cfg3.py r3 nospec spec profr3 profspec ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r4a_cfg3.txt; rc=${PIPESTATUS[0]}
cat gpurun_out/r4a_cfg3.txt; echo "cfg3 rc=$rc"
