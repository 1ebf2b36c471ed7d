#!/bin/bash
O=gpurun_out/r03c; mkdir -p $O; rm -f $O/*.json
timeout 1500 python -m pytest tests/test_gpu_config_scale.py -m gpu -x -q > $O/tests_seg.txt 2>&1; echo "tests rc=$?" >> $O/tests_seg.txt
tail -3 $O/tests_seg.txt
run() { name=$1; shift; env "$@" timeout 300 python bench.py --mode layer-split --steps 20 > $O/bench_layer_$name.json 2>> $O/err.txt; }
run default A=1
run elo_l1 GKR_DEBUG_SEG_ELO_MASK=0xff
run nomac GKR_DEBUG_SEG_NO_MAC=1
run nomac_l1 GKR_DEBUG_SEG_NO_MAC=1 GKR_DEBUG_SEG_ELO_MASK=0xff
run seg32 GKR_GATE_SEGMENT_LOG2=5
for f in $O/bench_layer_*.json; do python -c "
import json,sys
d=json.loads(open('$f').readline()); print('$f', round(d['ms_per_step'],3), d['matches_golden_digest'], d['kernel_ms_per_step'])"; done
