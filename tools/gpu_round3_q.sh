#!/bin/bash
O=gpurun_out/r03q; mkdir -p $O
run() { ki=$1; k=$2; shift; shift; env "$@" timeout 300 python bench.py --mode layer-split --k-i $ki --k $k --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(json.dumps({'k_i': $ki, 'k': $k, 'env': '$*', 'ms': round(d['ms_per_step'],3), 'ok': d['matches_golden_digest'], 'kernels': {a: round(b,4) for a,b in d['kernel_ms_per_step'].items()}}))" | tee -a $O/threshold.jsonl; }
for cfg in "16 8" "18 9" "19 9" "20 10" "22 11"; do
  set -- $cfg
  run $1 $2 GKR_GATE_SEGMENTS_MIN_LOG2=16
  run $1 $2 GKR_GATE_SEGMENTS_OFF=1
done
