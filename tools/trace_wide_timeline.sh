# kernel timeline of the last layer sumcheck of tools/bench_wide.py at one shape (default 20,20): start, duration, gap to the predecessor
R=${GRAFT_REPO_ROOT:-/root/repo}
SHAPE=${1:-20,20}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trw
rocprofv3 --kernel-trace --output-format csv -d /tmp/trw -- python3 $R/tools/bench_wide.py $SHAPE > /tmp/trw.out 2>&1
F=$(ls /tmp/trw/*/*kernel_trace.csv | head -1)
python3 - "$F" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last sumcheck: from the last k_layer_prologue on
idx=max(i for i,r in enumerate(rows) if 'k_layer_prologue' in r['Kernel_Name'])
last=rows[idx-1:]
t0=int(last[0]['Start_Timestamp'])
prev_end=None
for r in last:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    gap=(s-prev_end)/1000 if prev_end else 0
    print("%9.1f us  dur %7.1f  gap %7.1f  %s"%((s-t0)/1000,(e-s)/1000,gap,r['Kernel_Name'][:70]))
    prev_end=e
P
