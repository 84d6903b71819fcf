# development aid: parity tests of the BAND kernel with the shipped build, then variants on ONE box
# usage: gpurun -- 'bash scripts/gpu_ab.sh "head base" "profbase" "prof2"'   (timing tags, -DSMM_PROFILE tags, -DSMM_PROFILE=2 tags)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_viterbi.py -q -x -k "band or structured" > gpurun_out/ab_tests.txt 2>&1; tail -2 gpurun_out/ab_tests.txt
: > gpurun_out/ab.txt
[ -n "$1" ] && timeout -k 10 300 python scripts/time_variants.py $1 >> gpurun_out/ab.txt 2>&1
[ -n "$1" ] && timeout -k 10 300 python scripts/prof_cfg3.py $1 >> gpurun_out/ab.txt 2>&1
[ -n "$2" ] && timeout -k 10 300 python scripts/prof_cfg3.py $2 >> gpurun_out/ab.txt 2>&1
[ -n "$3" ] && SMM_PROF_LAST=1 timeout -k 10 300 python scripts/prof_cfg3.py $3 >> gpurun_out/ab.txt 2>&1
echo rc=$?
