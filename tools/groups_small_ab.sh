#!/bin/bash
# A/B: the lane-group gate passes (kernels_wide.hip) forced onto the circom-sized layers of configs[0] / configs[3]
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in 13 1; do
  for rep in 1 2; do
    GKR_GATE_GROUPS_MIN_K=$v GKR_BENCH_NO_VERIFY=1 python3 $R/bench.py --mode proofs --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('min_k=$v configs[3] ms/step %.3f' % d['ms_per_step'])"
  done
  GKR_GATE_GROUPS_MIN_K=$v python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['aggregated_proofs']; print('min_k=$v config0 ms %.3f %s config3 %.3f verified %s' % (a['config0_three_inputs']['ms'], a['config0_three_inputs']['ms_each'], a['config3']['ms'], a['verified_ok']))"
done
