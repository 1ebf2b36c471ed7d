#!/bin/bash
O=gpurun_out/r03i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_config_scale.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python tools/bench_layer.py --k-i 24 --k 12 --steps 3 > $O/bench_layer_one_shot.json 2>$O/err2.txt
python -c "
import json
d=json.loads(open('$O/bench_layer_one_shot.json').readline()); print(d['wall_ms_per_sumcheck'], {k:round(v,3) for k,v in d['kernel_ms_per_sumcheck'].items() if v}, d['matches_golden_digest'])"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l -- python3 $GRAFT_REPO_ROOT/tools/bench_layer.py --k-i 24 --k 12 --steps 3 > /dev/null 2>&1
cp $(ls /tmp/prof_l/*/*kernel_stats.csv | head -1) $GRAFT_REPO_ROOT/$O/kernel_stats_layer_one_shot.csv; head -16 $GRAFT_REPO_ROOT/$O/kernel_stats_layer_one_shot.csv | cut -d, -f1-6
