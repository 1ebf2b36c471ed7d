#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03r; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_config_scale.py -m gpu -x -q -k "config5_layer_every_form or scenarios" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l -- python3 $GRAFT_REPO_ROOT/bench.py --mode layer-split --steps 10 --warmup 2 > $O/line.json 2>/dev/null
cp $(ls /tmp/prof_l/*/*kernel_stats.csv | head -1) $O/kernel_stats_layer_split.csv
python3 - <<'PY'
import csv,os,json
O=os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out/r03r')
for r in list(csv.DictReader(open(os.path.join(O,'kernel_stats_layer_split.csv'))))[:12]:
    print('  ',r['Name'][:50].ljust(50), r['Calls'], round(float(r['AverageNs'])/1e3,1))
d=json.loads(open(os.path.join(O,'line.json')).readline()); print(d['ms_per_step'], d['matches_golden_digest'])
PY
