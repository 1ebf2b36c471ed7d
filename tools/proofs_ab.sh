#!/bin/bash
# A/B of gkr_prove_many's item splitting on configs[3] (64 inputs x 12 sub-circuits): ms per step for several piece counts
R=${GRAFT_REPO_ROOT:-/root/repo}
for p in 0 14 16 19 24; do
  for rep in 1 2; do
    GKR_PROVE_MANY_PIECES=$p GKR_BENCH_NO_VERIFY=1 python3 $R/bench.py --mode proofs --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pieces', $p, 'ms/step %.3f' % d['ms_per_step'], 'proofs/s %.0f' % d['value'])"
  done
done
