#!/bin/bash
O=gpurun_out/r03f; mkdir -p $O
one() { env "$@" timeout 300 python bench.py --n 16 --batch 4096 --no-extras --proofs 0 --no-cpu-baseline --no-verify --steps 10 --warmup 3 --layer-k-i 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(json.dumps({'env': '$*', 'ms': round(d['ms_per_step'],3), 'value': d['value'], 'kernel_ms': d['kernel_ms']}))" | tee -a $O/n16_groups2.jsonl; }
one GKR_GROUP_SIZE=256 GKR_PASS_QUEUE_DEPTH=4
one GKR_GROUP_SIZE=256 GKR_PASS_QUEUE_DEPTH=8
one GKR_GROUP_SIZE=128 GKR_PASS_QUEUE_DEPTH=4
one GKR_GROUP_SIZE=128 GKR_PASS_QUEUE_DEPTH=8
one GKR_GROUP_SIZE=128 GKR_PASS_QUEUE_DEPTH=16
one GKR_GROUP_SIZE=256 GKR_PASS_QUEUE_DEPTH=4 GKR_ROUNDS_PER_PASS=4
one GKR_GROUP_SIZE=128 GKR_PASS_QUEUE_DEPTH=8 GKR_ROUNDS_PER_PASS=4
one GKR_GROUP_SIZE=256 GKR_PASS_QUEUE_DEPTH=4 GKR_HASH_CHUNK=8
one GKR_GROUP_SIZE=256 GKR_PASS_QUEUE_DEPTH=4 GKR_HASH_CHUNK=16
