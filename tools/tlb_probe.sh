# UTCL1 (address-translation cache) counters of the fold pass, per table placement: does a slow placement miss more?
#   bash tools/tlb_probe.sh      (on the GPU box; prints the probe's bandwidth per trial and the counters per trial)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tlbp
rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum --output-format csv -d /tmp/tlbp -- python3 $GRAFT_REPO_ROOT/tools/alloc_mode_probe.py 128 > /tmp/tlbp_out.txt 2>&1
cat /tmp/tlbp_out.txt | grep trial
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/tlbp/**/*counter_collection.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_mle_multifold_mfma<5>' in r['Kernel_Name']]
per = collections.OrderedDict()
for r in rows:
    per.setdefault(r['Dispatch_Id'], {})[r['Counter_Name']] = float(r['Counter_Value'])
d = list(per.values())
print(len(d), 'dispatches of the <5> pass')
per_trial = len(d) // 8
for t in range(8):
    chunk = d[t * per_trial:(t + 1) * per_trial]
    miss = sum(x.get('TCP_UTCL1_TRANSLATION_MISS_sum', 0) for x in chunk) / len(chunk)
    hit = sum(x.get('TCP_UTCL1_TRANSLATION_HIT_sum', 0) for x in chunk) / len(chunk)
    print('trial %d: UTCL1 misses per launch %.3e, hits %.3e, miss rate %.4f' % (t, miss, hit, miss / (miss + hit) if miss + hit else 0))
PY
