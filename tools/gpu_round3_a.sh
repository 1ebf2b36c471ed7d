#!/bin/bash
# first GPU pass of round 3: new parity tests, the default bench line, the one-rank RCCL exchange
O=gpurun_out/r03a; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_parity.py -m gpu -x -q -k "device_exchange or headline or gate_sharded or resident" > $O/tests_new.txt 2>&1; echo "tests rc=$?" >> $O/tests_new.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/bench_default.err
GKR_BENCH_FORCE_GROUP=1 timeout 300 python bench.py --mode layer-split --steps 20 > $O/bench_layer_split_rccl1.json 2> $O/bench_layer_split_rccl1.err; echo "rc=$?" >> $O/bench_layer_split_rccl1.err
timeout 300 python bench.py --mode layer-split --steps 20 > $O/bench_layer_split.json 2> $O/bench_layer_split.err
tail -3 $O/tests_new.txt; tail -2 $O/bench_default.err; head -c 1500 $O/bench_layer_split_rccl1.json; tail -3 $O/bench_layer_split_rccl1.err
