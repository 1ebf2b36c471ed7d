# same box: the library against variants built by tools/build_variant.sh; prints the gate passes' kernel times of configs[4]
for v in "" "$@"; do
  if [ -n "$v" ]; then export GKR_AMD_LIB=$PWD/tools/_variants/$v/libgkr_amd.so GKR_EXPERIMENT_WRONG_RESULTS_OK=1; fi
  python tools/bench_layer.py --resident --steps 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_sumcheck']
print('${v:-baseline}', 'wall %.3f' % d['wall_ms_per_sumcheck'], 'gate_uv %.3f gate_rows %.3f' % (k['gate_uv'], k['gate_rows']), d.get('matches_golden_digest'))"
done
