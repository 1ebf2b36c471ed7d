#!/bin/bash
# A/B of the product passes' fused publish (the last block of a pass totals and publishes) and the adaptive heavy-bucket units
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "" 1; do
  echo "GKR_NO_FUSED_PUBLISH=[$v]"
  GKR_NO_FUSED_PUBLISH=$v python3 $R/tools/bench_large_r1cs.py 30 14
  GKR_NO_FUSED_PUBLISH=$v python3 $R/bench.py --mode layer-split --steps 30 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' layer24 ms/step %.4f golden %s' % (d['ms_per_step'], d['matches_golden_digest']))"
  for rep in 1 2; do
  GKR_NO_FUSED_PUBLISH=$v GKR_BENCH_LARGE_R1CS=0 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['aggregated_proofs']; print(' config0 ms %.3f  config3 ms %.3f %s' % (a['config0_three_inputs']['ms'], a['config3']['ms'], a['verified_ok']))"
  done
  GKR_NO_FUSED_PUBLISH=$v python3 $R/tools/bench_wide_prove.py 18,20,20 | cut -c1-120
done
