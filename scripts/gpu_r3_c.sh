set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
(cat /sys/fs/cgroup/cpu.max; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us; nproc; python -c "import os;print(len(os.sched_getaffinity(0)))") > gpurun_out/r3c_host.txt 2>&1
cat gpurun_out/r3c_host.txt
timeout -k 10 1000 python -m pytest tests -q -m gpu --durations=25 > gpurun_out/r3c_pytest.log 2>&1 ; echo "tests rc=$?"
tail -45 gpurun_out/r3c_pytest.log
timeout -k 10 200 python scripts/probe_predict_cfg4.py > gpurun_out/r3c_probe_cfg4.txt 2>&1 ; echo "probe rc=$?"
grep -A12 "^--- pass\|^warm-up\|^fused\|^per batch" gpurun_out/r3c_probe_cfg4.txt | head -50
