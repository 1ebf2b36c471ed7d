#!/bin/bash
# Same-box A/B of two source states: builds a VARIANT of the library from a patched copy of gkr_amd/csrc into
# tools/_variants/<name>/libgkr_amd.so (git-ignored; travels to the GPU box); run with GKR_AMD_LIB=<that file>.
#   tools/build_variant.sh <name> '<sed script applied to kernels.hip>' [file]
set -e
NAME=$1; SED=$2; FILE=${3:-kernels.hip}
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/repo/gkr_amd $T/repo/include
cp -r $R/gkr_amd/csrc $T/repo/gkr_amd/csrc
cp $R/include/gkr_amd.h $T/repo/include/
sed -i "$SED" $T/repo/gkr_amd/csrc/$FILE
if diff -q $T/repo/gkr_amd/csrc/$FILE $R/gkr_amd/csrc/$FILE > /dev/null; then echo "the sed script changed nothing"; exit 1; fi
make -C $T/repo/gkr_amd/csrc -j4 > $T/make.log 2>&1 || { tail -30 $T/make.log; exit 1; }
mkdir -p $R/tools/_variants/$NAME
cp $T/repo/gkr_amd/lib/libgkr_amd.so $R/tools/_variants/$NAME/libgkr_amd.so
rm -rf $T
echo built tools/_variants/$NAME/libgkr_amd.so
