# round 4: BAND-only variants: parity tests of the K > 512 kernels through the FIRST tag, then timings of all (probe lattices + cfg3 DP kernel)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
first=${1%% *}
SMM_LIB_PATH=$PWD/action-segmentation_amd/libsmmdp_$first.so timeout -k 10 600 python -m pytest tests/test_gpu_viterbi.py -m gpu -q -x -k "band_mode or long_segment or 24_to_32 or 22_23 or masked or ramps or one_parameter" > gpurun_out/r4ab_tests.txt 2>&1; rc=$?
tail -3 gpurun_out/r4ab_tests.txt; echo "tests rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
( timeout -k 10 300 python scripts/time_variants.py $1
  timeout -k 10 300 python scripts/prof_cfg3.py $1 ) 2>&1 | grep -v "amdgpu.ids\|pass 0" > gpurun_out/r4ab.txt
cat gpurun_out/r4ab.txt
