#!/bin/bash
# A/B of the launch-ahead of the layer's product passes: the layer24 leg and the three-input proving step, on / off
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "" 1; do
  for rep in 1 2; do
    GKR_NO_LAUNCH_AHEAD=$v python3 $R/bench.py --mode layer-split --steps 30 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_ahead=[$v] layer24 ms/step %.4f golden %s' % (d['ms_per_step'], d['matches_golden_digest']))"
  done
done
for v in "" 1; do
  GKR_NO_LAUNCH_AHEAD=$v python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['aggregated_proofs']; print('no_ahead=[$v] config0 ms %.3f %s  config3 ms %.3f' % (a['config0_three_inputs']['ms'], a['config0_three_inputs']['ms_each'], a['config3']['ms']), a['verified_ok'])"
done
