#!/bin/bash
O=gpurun_out/r03j; mkdir -p $O
GKR_DEBUG_TIMING=1 timeout 300 python bench.py --no-extras --proofs 0 --no-cpu-baseline --no-verify --steps 6 --warmup 2 --layer-k-i 0 2> $O/timing.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms'], d['step_ms_each'])"
grep "gkr timing" $O/timing.txt | tail -12
