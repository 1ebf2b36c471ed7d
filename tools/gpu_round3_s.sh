#!/bin/bash
O=gpurun_out/r03s; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_scale.py tests/test_gpu_circom_pipeline.py -m gpu -x -q 2>&1 | tail -3
run() { name=$1; shift; env "$@" timeout 300 python bench.py --mode layer-split --steps 30 > $O/bench_layer_$name.json 2>> $O/err.txt; }
run fused A=1
run two_kernels GKR_PROD_PUBLISH_KERNEL=1
run fused2 A=1
run two_kernels2 GKR_PROD_PUBLISH_KERNEL=1
for f in $O/bench_layer_*.json; do python -c "
import json,sys
d=json.loads(open('$f').readline()); print('$f', round(d['ms_per_step'],3), d['matches_golden_digest'], d['kernel_ms_per_step'])"; done
