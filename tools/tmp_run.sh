#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_circom_pipeline.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --mode layer-split --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],3), d['matches_golden_digest'])"; done
PROBE_REPS=40 PROBE_THREADS=14 python tools/proof_many_probe.py 64 2>/dev/null | tail -1
PROBE_REPS=40 PROBE_THREADS=14 python tools/proof_many_probe.py 3 2>/dev/null | tail -1
