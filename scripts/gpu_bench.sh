set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 5 --warmup 2 2>&1 | tail -1 | tee gpurun_out/bench_cfg3.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg3 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_cfg3.log 2>&1
find gpurun_out/prof_cfg3 -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-160 | head -4
SMM_DEBUG_FLAGS=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('fwd only', r['roofline']['kernel_ms'], r['ms_per_step'])"
