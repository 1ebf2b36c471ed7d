#!/bin/bash
# ThreadSanitizer build of the library's HOST side (CPU only; never run on the GPU box -- listed in .gpurunignore):
# the C ABI's translation units' (gkr_capi.hip, capi_mle / _layer / _prove.hip) host passes instrumented (device code compiled as usual, nothing here needs a GPU at run time), the host-only
# units, and tests/first_use_race.cpp linked against it.  One sanitizer runtime for all of it: the ROCm clang's (hipcc
# instruments with it; gcc's libtsan lacks its entry points).   bash tests/build_host_tsan.sh  ->  gkr_amd/build_san/first_use_race_tsan
set -e
cd "$(dirname "$0")/../gkr_amd/csrc"
make -s kernels.o kernels_wide.o kernels_layer_dense.o exchange_rccl.o
OUT=../build_san
mkdir -p $OUT
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
CLANGXX=${CLANGXX:-/opt/rocm/lib/llvm/bin/clang++}
SAN="-fsanitize=thread"
for u in gkr_capi capi_mle capi_layer capi_prove; do
  $HIPCC --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Wno-unused-function -Xarch_host $SAN -c $u.hip -o $OUT/${u}_tsan.o
done
$CLANGXX -O1 -g -std=c++17 -fPIC $SAN -Wno-unknown-pragmas -mavx512f -mavx512ifma -mavx512vl -c mimc_ifma.cpp -o $OUT/mimc_ifma_tsan.o
for u in keccak circom_input r1cs mimc_adx; do $CLANGXX -O1 -g -std=c++17 -fPIC $SAN -Wno-unknown-pragmas -c $u.cpp -o $OUT/${u}_tsan.o; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $SAN -o $OUT/libgkr_tsan.so $OUT/gkr_capi_tsan.o $OUT/capi_mle_tsan.o $OUT/capi_layer_tsan.o $OUT/capi_prove_tsan.o kernels.o kernels_wide.o kernels_layer_dense.o exchange_rccl.o -ldl $OUT/mimc_ifma_tsan.o $OUT/keccak_tsan.o \
    $OUT/circom_input_tsan.o $OUT/r1cs_tsan.o $OUT/mimc_adx_tsan.o
$CLANGXX -O1 -g -std=c++17 $SAN -pthread ../../tests/first_use_race.cpp -L$OUT -lgkr_tsan -Wl,-rpath,'$ORIGIN' -o $OUT/first_use_race_tsan
