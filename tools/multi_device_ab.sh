# Same box, same minute: configs[3] through gkr_ctx_create_multi (bench.py --mode multi-device, standalone, no parent process
# holding a context) against the direct path (--mode proofs) -- where do the 0.9 ms of the default line's multi_device leg go?
for i in 1 2 3; do
  python bench.py --mode multi-device --proofs 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('multi-device standalone', d['ms'], d['ms_each'])"
  GKR_BENCH_DETAIL=/tmp/p.json python bench.py --mode proofs --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('proofs mode', d['ms_per_step'])"
done
