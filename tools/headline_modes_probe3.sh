# on whatever box this lands: the headline's fold-pass fraction under the launch-geometry switches -- if the box is in the slow mode
# (frac ~0.75), does any of them recover the fast one (~0.80)?
run() {
  GKR_BENCH_DETAIL=/tmp/hm.json env "$@" python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-verify --proofs 0 > /dev/null 2>&1
  python - "$*" <<'P'
import json,sys
d=json.load(open('/tmp/hm.json')); r=d['roofline']; f=r.get('first_fold_pass_GBps',{})
print('%-28s'%sys.argv[1], 'ms %.3f'%d['ms_per_step'], 'frac %.4f'%r['frac'], 'fold min/med/max %.0f %.0f %.0f'%(f.get('min',0),f.get('median',0),f.get('max',0)), 'first_pass %.0f'%r.get('first_pass_GBps',0))
P
}
run X=1
if [ "${PROBE_ONLY_SLOW:-0}" = "1" ] && python -c "
import json,sys; sys.exit(0 if json.load(open('/tmp/hm.json'))['roofline']['frac'] >= 0.775 else 1)"; then echo "fast box: nothing to learn here"; exit 0; fi
run X=1
run GKR_FOLD_MIN_CHUNK=512
run GKR_FOLD_MIN_CHUNK=128
run GKR_FOLD_BLOCKS=16384
run GKR_FOLD_BLOCKS=262144
run GKR_GROUP_SIZE=64
run GKR_GROUP_SIZE=256
run GKR_NO_LATE_STREAM=1
run X=1
