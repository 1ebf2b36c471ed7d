# same box: tools/bench_wide.py under the library and under variants built by tools/build_variant.sh
#   bash tools/wide_variants_ab.sh "20,20 20,15" eqsplit ...
SHAPES=$1; shift
for v in "" "$@"; do
  if [ -n "$v" ]; then export GKR_AMD_LIB=$PWD/tools/_variants/$v/libgkr_amd.so; fi
  echo "== ${v:-baseline}"
  python tools/bench_wide.py $SHAPES 2>&1 | tail -n $(echo $SHAPES | wc -w)
  WIDE_SHAPE=circom python tools/bench_wide.py 20,20 2>&1 | tail -1
done
