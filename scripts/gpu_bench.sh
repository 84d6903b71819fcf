set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_viterbi.py -x -q -m gpu -k emission 2>&1 | tail -2
python bench.py --steps 5 --warmup 2 2>&1 | tail -1 | tee gpurun_out/bench_cfg3.json | cut -c1-400
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg3 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_cfg3.log 2>&1
find gpurun_out/prof_cfg3 -name "*kernel_stats.csv" | sort | tail -1 | xargs cat | cut -c1-160 | head -4
