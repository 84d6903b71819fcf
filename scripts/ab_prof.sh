# same-box rocprofv3 A/B of builds of libsmmdp.so: per-kernel averages of the bench.
#   bash scripts/ab_prof.sh <workload> <kernel name pattern> lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
W=${1:-cfg3}; PAT=${2:-smm_emission_kernel}; shift 2
for rep in 1 2; do
for lib in "$@"; do
  rm -rf gpurun_out/ab_$lib
  SMM_LIB_PATH=$GRAFT_REPO_ROOT/action-segmentation_amd/$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$lib -- python bench.py --workload $W --steps 8 --warmup 2 --no-cpu-baseline --no-predict-e2e --no-strong-leg --second-seed -1 > gpurun_out/ab_$lib.log 2>&1
  echo "== $W $lib"
  grep -h "$PAT" gpurun_out/ab_$lib/*/*kernel_stats.csv | awk -F'",' '{print substr($1,1,50), $2}'
done
done
