# Kernel timeline of ONE 2^20 sumcheck (batch 1, host transcript) on the GPU box:  bash tools/trace_mle_latency_timeline.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/trm -- python3 $R/bench.py --steps 3 --warmup 1 --batch 1 --no-cpu-baseline --no-verify --no-extras --proofs 0 --layer-k-i 0 > /tmp/trm.json 2>/dev/null
F=$(ls /tmp/trm/*/*kernel_trace.csv | head -1)
python3 - "$F" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the batch-1 latency leg runs after the timed steps: find the last run of kernels where k_mle_sub_sums grid is one table
start=max(0,len(rows)-30); end=len(rows)
t0=int(rows[start]['Start_Timestamp']); prev=None
for r in rows[start:end]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    print("%9.1f us  dur %7.1f  gap %7.1f  %s"%((s-t0)/1000,(e-s)/1000,((s-prev)/1000 if prev else 0),r['Kernel_Name'][:70]))
    prev=e
P
