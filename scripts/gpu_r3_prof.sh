set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
(cd action-segmentation_amd/csrc && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -DSMM_PROFILE -DSMM_DEV_R=16 $SMM_PROF_FLAGS -mllvm -pragma-unroll-threshold=1048576 -mllvm -unroll-threshold=1048576 -o ../libsmmdp_prof16.so smm_api.hip smm_emission.hip smm_viterbi.hip smm_logz.hip smm_logz_bwd.hip smm_dense.hip smm_eval.hip smm_fit.hip smm_tables.hip)
timeout -k 10 300 python scripts/prof_band.py prof plain > gpurun_out/r3_band_stamps.txt 2>&1; echo rc=$?
cat gpurun_out/r3_band_stamps.txt
