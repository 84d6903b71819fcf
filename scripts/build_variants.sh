# development aid: libsmmdp_<tag>.so = the shipped objects + smm_viterbi.hip compiled with extra flags, BAND kernels only
# usage: bash scripts/build_variants.sh tag1 "-DSMM_ABLATE=1" tag2 "-DSMM_ABLATE=3" ...   (objects of the normal build must exist)
set -e
cd "$(dirname "$0")/../action-segmentation_amd/csrc"
F="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -mllvm -pragma-unroll-threshold=1048576 -mllvm -unroll-threshold=1048576 ${SMM_VARIANT_NAN--fno-honor-nans -DSMM_FMAX_BUILTIN}"
pids=""
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  src=${SMM_VARIANT_SRC:-smm_viterbi}     # the translation unit that is recompiled (smm_viterbi, smm_logz, ...)
  ( hipcc $F ${SMM_VARIANT_FULL:--DSMM_DEV_BAND_ONLY} $flags -c -o _obj/${src}_$tag.o $src.hip &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsmmdp_$tag.so $(for u in smm_api smm_emission smm_viterbi smm_chunk smm_logz smm_logz_bwd smm_dense smm_eval smm_fit smm_tables; do [ $u != $src ] && echo _obj/$u.o; done) _obj/${src}_$tag.o && echo built $tag ) &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
