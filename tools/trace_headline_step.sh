# kernel timeline of the LAST step of the headline workload (1024 x 2^20): every kernel of >= 100 us with its stream, and the
# intervals in which no kernel at all was running -- where a 12 ms step is not its 11.4 ms of streaming kernels
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trh
rocprofv3 --kernel-trace --output-format csv -d /tmp/trh -- python3 $R/bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-verify --proofs 0 > /tmp/trh.out 2>/dev/null
F=$(find /tmp/trh -name '*kernel_trace.csv' | head -1)
python3 - "$F" <<'P'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'fill_table' not in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# steps: split at gaps > 2 ms?  take the kernels after the last k_mle_sub_sums run of 8 launches: simpler -- the last 1/4 of the trace
subs=[i for i,r in enumerate(rows) if 'k_mle_sub_sums' in r['Kernel_Name']]
first=subs[-8]
last=rows[first:]
t0=int(last[0]['Start_Timestamp'])
busy_until=t0; idle=[]; 
for r in last:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    if s>busy_until: idle.append((busy_until-t0, s-busy_until))
    busy_until=max(busy_until,e)
    d=(e-s)/1000
    if d>=100: print("%9.1f us  dur %7.1f  q%s  %s" % ((s-t0)/1000, d, r.get('Queue_Id','?'), r['Kernel_Name'].split('(')[0][-40:]))
print("step span %.1f us, idle total %.1f us in %d gaps; gaps >= 20 us:" % ((busy_until-t0)/1000, sum(g for _,g in idle)/1000, len(idle)))
for at,g in idle:
    if g>=20000: print("   at %9.1f us: %.1f us" % (at/1000, g/1000))
P
