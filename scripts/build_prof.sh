# diagnostic library with in-kernel s_memtime stamps (never shipped / never timed)
cd "$(dirname "$0")/../action-segmentation_amd/csrc" && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -DSMM_PROFILE -mllvm -pragma-unroll-threshold=1048576 -mllvm -unroll-threshold=1048576 -o ../libsmmdp_prof.so smm_api.hip smm_emission.hip smm_viterbi.hip smm_logz.hip smm_logz_bwd.hip
