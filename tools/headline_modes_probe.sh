# same box, fresh process each time: the headline's step time, the fold pass's fraction of peak, where the tables landed and the
# first-fold launches' rates (min / median / max) -- does a process draw a "mode"?
for i in $(seq 1 ${1:-6}); do
  GKR_BENCH_DETAIL=/tmp/hm_$i.json python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-verify --proofs 0 $2 > /dev/null 2>&1
  python - $i <<'P'
import json,sys
d=json.load(open('/tmp/hm_%s.json'%sys.argv[1])); r=d['roofline']; f=r.get('first_fold_pass_GBps',{})
print(sys.argv[1], 'ms %.3f'%d['ms_per_step'], 'frac %.4f'%r['frac'], 'addr', d['tables_device_address'], 'fold GB/s min/med/max %.0f %.0f %.0f'%(f.get('min',0),f.get('median',0),f.get('max',0)), 'unshared %.0f'%f.get('unshared_median',0), 'first_pass %.0f'%r.get('first_pass_GBps',0), 'copy %.0f'%(r.get('copy_GBps_measured') or 0))
P
done
