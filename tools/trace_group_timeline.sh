# kernel timeline of ONE lockstep group of the 262 144-constraint step (tools/bench_large_r1cs.py): the kernels whose grid has
# BATCH proofs in y (default 7: the seven deep sub-circuits), last step only: start, duration, gap to the group's previous kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
BATCH=${1:-7}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trg
rocprofv3 --kernel-trace --output-format csv -d /tmp/trg -- python3 $R/tools/bench_large_r1cs.py 3 14 > /tmp/trg.out 2>&1
F=$(ls /tmp/trg/*/*kernel_trace.csv | head -1)
python3 - "$F" "$BATCH" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
B=int(sys.argv[2])
rows.sort(key=lambda r:int(r['Start_Timestamp']))
def gy(r):
    return int(r.get('Grid_Size_Y', r.get('Grid_Size_y', 0)) or 0) // max(1, int(r.get('Workgroup_Size_Y', r.get('Workgroup_Size_y', 1)) or 1))
mine=[r for r in rows if gy(r)==B]
# the last step: from the last k_layer_eval of this batch on (five per step: go back to the first of the last five)
ev=[i for i,r in enumerate(mine) if 'k_layer_eval' in r['Kernel_Name']]
start=ev[-5] if len(ev)>=5 else 0
last=mine[start:]
t0=int(last[0]['Start_Timestamp']); prev=None
tot_k=0; tot_gap=0
for r in last:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    gap=(s-prev)/1000 if prev else 0
    tot_k+=(e-s)/1000; tot_gap+=max(0,gap)
    print("%9.1f us  dur %7.1f  gap %7.1f  %s"%((s-t0)/1000,(e-s)/1000,gap,r['Kernel_Name'].replace('void ','').replace('gkr::','')[:48]))
    prev=e
print("kernels %.0f us, gaps %.0f us, launches %d" % (tot_k, tot_gap, len(last)))
P
