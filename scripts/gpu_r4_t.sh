# round 4: the GPU suite (+ optional extra command)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r4_tests.txt 2>&1; rc=$?
tail -8 gpurun_out/r4_tests.txt; echo "tests rc=$rc"
