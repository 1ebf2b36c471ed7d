#!/bin/bash
# segment gate passes: parity at every width, then the layer bench in its variants
O=gpurun_out/r03b; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_config_scale.py tests/test_gpu_sharded.py -m gpu -x -q > $O/tests_seg.txt 2>&1; echo "tests rc=$?" >> $O/tests_seg.txt
run() { name=$1; shift; env "$@" timeout 300 python bench.py --mode layer-split --steps 20 > $O/bench_layer_$name.json 2>> $O/bench_layer_split.err; }
run default A=1
run seg32 GKR_GATE_SEGMENT_LOG2=5
run seg8 GKR_GATE_SEGMENT_LOG2=3
run nolds GKR_GATE_SEGMENTS_NO_LDS=1
run off GKR_GATE_SEGMENTS_OFF=1
tail -5 $O/tests_seg.txt
for f in $O/bench_layer_*.json; do python -c "
import json,sys
d=json.loads(open('$f').readline()); print('$f', round(d['ms_per_step'],3), d['matches_golden_digest'], d['kernel_ms_per_step'], d['roofline']['frac'])"; done
