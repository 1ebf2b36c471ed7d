#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03l; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l -- python3 $GRAFT_REPO_ROOT/bench.py --mode layer-split --steps 10 --warmup 2 > /dev/null 2>&1
cp $(ls /tmp/prof_l/*/*kernel_stats.csv | head -1) $O/kernel_stats_layer_split.csv
GKR_GATE_SEGMENT_LOG2=5 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l5 -- python3 $GRAFT_REPO_ROOT/bench.py --mode layer-split --steps 10 --warmup 2 > /dev/null 2>&1
cp $(ls /tmp/prof_l5/*/*kernel_stats.csv | head -1) $O/kernel_stats_layer_split_seglog5.csv
python3 - <<'PY'
import csv,os
for f in ('kernel_stats_layer_split.csv','kernel_stats_layer_split_seglog5.csv'):
    print(f)
    for r in list(csv.DictReader(open(os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out/r03l',f))))[:9]:
        print('  ',r['Name'][:50].ljust(50), r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
