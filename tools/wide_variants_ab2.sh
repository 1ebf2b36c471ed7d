SHAPES=$1; shift
for v in "" "$@"; do
  if [ -n "$v" ]; then export GKR_AMD_LIB=$PWD/tools/_variants/$v/libgkr_amd.so; fi
  echo "== ${v:-baseline}"
  python tools/bench_wide.py $SHAPES 2>&1 | tail -n $(echo $SHAPES | wc -w) | python -c "
import sys,ast
for l in sys.stdin:
    d=ast.literal_eval(l); k=d['kernel_ms_per_call']; print(d['k_i'],d['k'],'wall',min(d['resident_ms']),'gate+eq',round(k.get('eq_table_z',0)+k['gate_uv']+k['gate_rows'],4))"
done
