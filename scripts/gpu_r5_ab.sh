# round 5: same-box A/B of libsmmdp_<tag>.so variants (scripts/build_variants.sh): parity subset first, then lattices, cfg3 corpus, stamps
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${SMM_TAG:-r5_ab}
timeout -k 10 900 python -m pytest ${SMM_AB_TESTS:-tests/test_gpu_viterbi.py tests/test_gpu_fullsize.py} -q -m gpu -x > gpurun_out/${T}_pytest.log 2>&1 ; echo "tests rc=$?"
tail -5 gpurun_out/${T}_pytest.log
timeout -k 10 300 python scripts/time_variants.py ${SMM_AB_TIME:-base8 new} > gpurun_out/${T}_lattices.txt 2>&1
grep -v amdgpu.ids gpurun_out/${T}_lattices.txt
SMM_ONLY_BAND=1 timeout -k 10 300 python scripts/prof_cfg3.py ${SMM_AB_CFG3:-base8 new prof0 prof} > gpurun_out/${T}_cfg3.txt 2>&1
grep -v amdgpu.ids gpurun_out/${T}_cfg3.txt
SMM_PROF_LAST=1 timeout -k 10 300 python scripts/prof_cfg3.py ${SMM_AB_LAST:-profl} > gpurun_out/${T}_cfg3_last.txt 2>&1
grep -v amdgpu.ids gpurun_out/${T}_cfg3_last.txt
