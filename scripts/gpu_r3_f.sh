set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_eval.py -q -x > gpurun_out/r3f_pytest.log 2>&1 ; echo "tests rc=$?"
tail -8 gpurun_out/r3f_pytest.log
timeout -k 10 300 python scripts/probe_eval.py > gpurun_out/r3f_probe_eval.txt 2>&1; echo rc=$?
head -60 gpurun_out/r3f_probe_eval.txt
