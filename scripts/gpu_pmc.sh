# HBM traffic of the kernels from PMC counters (MI355X_MICROARCH.md §HBM: separate --pmc passes;
# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$c.log 2>&1
  ls gpurun_out/pmc_$c/*/ | head
done
python scripts/pmc_summary.py gpurun_out cfg3
