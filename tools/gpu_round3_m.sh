#!/bin/bash
O=gpurun_out/r03m; mkdir -p $O
one() { env "$@" timeout 300 python bench.py --proofs 0 --no-cpu-baseline --no-verify --steps 10 --warmup 3 --layer-k-i 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(json.dumps({'env': '$*', 'ms': round(d['ms_per_step'],3), 'lat1': round(d['latency_ms_batch1'],3), 'n16': round(d['n16']['ms_per_step'],3)}))" | tee -a $O/chunk_rule.jsonl; }
sed -i 's///' tools/gpu_round3_m.sh 2>/dev/null
for i in 1 2 3; do one A=1; one GKR_HASH_CHUNK=8; done
