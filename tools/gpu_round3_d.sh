#!/bin/bash
O=gpurun_out/r03d; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch or prove" > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/tests.txt; tail -3 $O/tests.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03d/bench_default.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['first_fold_pass_GBps'])
print('host_transcript', d['host_transcript'])
print('n16', {k:v for k,v in d['n16'].items() if k!='workload'})
print('layer24', d['layer24']['ms_per_step'], d['layer24']['roofline']['frac'], d['layer24']['kernel_ms_per_step'])
print('proofs', d['aggregated_proofs']['config3'])
print('verified', d['verified']['ok'])
PY
