# same box: what the HIP events of the roofline measurement cost the headline step (profile level 2 against none)
for i in 1 2 3; do
  for flag in "" "--no-profile"; do
    GKR_BENCH_DETAIL=/tmp/d.json python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-verify --proofs 0 $flag 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('${flag:-events on}', round(d['ms_per_step'],3), d['roofline'].get('frac'))"
  done
done
