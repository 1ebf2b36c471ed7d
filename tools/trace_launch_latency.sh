# launch-to-start latency of the kernels of ONE lockstep group of the 262 144-constraint step (grid.y = BATCH proofs; default 7: the deep
# group): for every kernel, (start on the device) - (return of its hipLaunchKernel call), and its duration -- with all three groups
# proving (contended) and what the distribution looks like.   bash tools/trace_launch_latency.sh [BATCH]
R=${GRAFT_REPO_ROOT:-/root/repo}
BATCH=${1:-7}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tll
rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d /tmp/tll -- python3 $R/tools/bench_large_r1cs.py 3 14 > /tmp/tll.out 2>&1
python3 - "$(find /tmp/tll -name '*kernel_trace.csv' | head -1)" "$(find /tmp/tll -name '*hip_api_trace.csv' | head -1)" "$BATCH" <<'P'
import csv,sys,statistics,collections
k=list(csv.DictReader(open(sys.argv[1])))
a=list(csv.DictReader(open(sys.argv[2])))
B=int(sys.argv[3])
api={}
for r in a:
    if 'Launch' in r.get('Function',''):
        api[r['Correlation_Id']]=(int(r['Start_Timestamp']),int(r['End_Timestamp']))
def gy(r):
    return int(r.get('Grid_Size_Y', 0) or 0)//max(1,int(r.get('Workgroup_Size_Y',1) or 1))
per=collections.defaultdict(list)
for r in k:
    if gy(r)!=B: continue
    c=r['Correlation_Id']
    if c not in api: continue
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    name=r['Kernel_Name'].replace('void ','').replace('gkr::','').split('(')[0]
    per[name].append(((s-api[c][1])/1000,(e-s)/1000,(api[c][1]-api[c][0])/1000))
print("%-34s %5s  %s"%("kernel (grid.y = %d)"%B,"n","launch call us (median) | call return -> start on the device us (median, p90) | duration us (median)"))
for n,v in sorted(per.items(), key=lambda kv:-len(kv[1])):
    lat=sorted(x[0] for x in v); dur=[x[1] for x in v]; call=[x[2] for x in v]
    print("%-34s %5d  %6.1f | %7.1f %7.1f | %7.1f"%(n[:34],len(v),statistics.median(call),statistics.median(lat),lat[int(0.9*(len(lat)-1))],statistics.median(dur)))
P
