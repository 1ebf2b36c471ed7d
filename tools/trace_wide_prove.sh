# kernel timeline of the LAST proof of tools/bench_wide_prove.py (a circuit with k = 18,20,20 by default): start, duration, gap, stream
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/twp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/twp -- python3 $R/tools/bench_wide_prove.py ${1:-18,20,20} > /tmp/twp.out 2>&1
tail -1 /tmp/twp.out
python3 - "$(find /tmp/twp -name '*kernel_trace.csv' | head -1)" "$(find /tmp/twp -name '*memory_copy_trace.csv' | head -1)" <<'P'
import csv,sys
rows=[dict(r, kind="k") for r in csv.DictReader(open(sys.argv[1]))]
try:
    for r in csv.DictReader(open(sys.argv[2])):
        rows.append({"Start_Timestamp": r["Start_Timestamp"], "End_Timestamp": r["End_Timestamp"], "Kernel_Name": "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "")), "Queue_Id": "-", "kind": "c"})
except Exception as e:
    print("no copy trace", e)
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_layer_eval' in r['Kernel_Name']]
# the last proof: its forward evaluation starts with the first of the last (layers) k_layer_eval launches
nl=2
start=idx[-nl] if len(idx)>=nl else 0
last=rows[start:]
t0=int(last[0]['Start_Timestamp']); prev=None
for r in last:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    gap=(s-prev)/1000 if prev else 0
    print("%9.1f us  dur %7.1f  gap %7.1f  q%-3s %s"%((s-t0)/1000,(e-s)/1000,gap,r.get('Queue_Id','?'),r['Kernel_Name'].replace('void ','').replace('gkr::','')[:60]))
    prev=max(prev or 0,e)
P
