# kernel timeline of the LAST sumcheck of tools/bench_wide.py at a shape (default 20,20): start, duration, gap to the previous kernel
#   bash tools/trace_wide_timeline.sh 20,15 > gpurun_out/wide_timeline_20_15.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
SHAPE=${1:-20,20}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trw
rocprofv3 --kernel-trace --output-format csv -d /tmp/trw -- python3 $R/tools/bench_wide.py $SHAPE > /tmp/trw.out 2>&1
tail -1 /tmp/trw.out | cut -c1-200
F=$(find /tmp/trw -name '*kernel_trace.csv' | head -1)
python3 - "$F" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last sumcheck: from the last k_layer_prologue on
idx=[i for i,r in enumerate(rows) if 'k_layer_prologue' in r['Kernel_Name']]
last=rows[idx[-1]:] if idx else rows[-80:]
t0=int(last[0]['Start_Timestamp']); prev=None; tk=0; tg=0
for r in last:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    gap=(s-prev)/1000 if prev else 0
    tk+=(e-s)/1000; tg+=max(gap,0)
    print("%9.1f us  dur %7.1f  gap %7.1f  %s"%((s-t0)/1000,(e-s)/1000,gap,r['Kernel_Name'].replace('void ','').replace('gkr::','')[:70]))
    prev=e
print("kernels %.0f us, gaps %.0f us, launches %d"%(tk,tg,len(last)))
P
