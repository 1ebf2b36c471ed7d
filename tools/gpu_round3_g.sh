#!/bin/bash
O=gpurun_out/r03g; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
GKR_BENCH_BACKEND=gloo GKR_BENCH_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --batch 256 --proofs 16 > $O/bench_2ranks_one_gpu_gloo.json 2> $O/bench_2ranks.err; echo "rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03g/bench_2ranks_one_gpu_gloo.json').readline())
print(d['n_gpus'], d['ms_per_step'], d['verified']['ok'], d['verified'].get('all_ranks_ok'))
print('layer24', d['layer24']['ms_per_step'], d['layer24']['matches_golden_digest'])
print('split', json.dumps(d.get('layer24_split'))[:600])
PY
tail -5 $O/bench_2ranks.err
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03g/bench_default.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['first_fold_pass_GBps'])
print('n16', d['n16']['ms_per_step'], d['n16']['value'], d['n16']['roofline']['frac'], d['n16']['whole_batch_digest'])
PY
