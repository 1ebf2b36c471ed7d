R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 $R/bench.py --mode layer-split --steps 4 --warmup 2 > /tmp/tr.json 2>/dev/null
F=$(ls /tmp/tr/*/*kernel_trace.csv | head -1)
python3 - "$F" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step: find last occurrences; print the final 60 kernels with gaps
last=rows[-70:]
t0=int(last[0]['Start_Timestamp'])
prev_end=None
for r in last:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    gap=(s-prev_end)/1000 if prev_end else 0
    print("%9.1f us  dur %7.1f  gap %7.1f  %s"%((s-t0)/1000,(e-s)/1000,gap,r['Kernel_Name'][:60]))
    prev_end=e
P
