#!/bin/bash
# rocprofv3 kernel stats of tools/bench_large_r1cs.py (the 262 144-constraint step through gkr_prove_many) ->
# gpurun_out/<tag>_kernel_stats.csv and a per-kernel summary on stdout.  usage: tools/stats_large_r1cs.sh <tag> [reps] [threads]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-large_r1cs}; REPS=${2:-10}; THREADS=${3:-14}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lr -- python3 $R/tools/bench_large_r1cs.py $REPS $THREADS > /tmp/lr.out 2>&1
f=$(find /tmp/lr -name "*kernel_stats.csv" | head -1)
mkdir -p $R/gpurun_out
cp "$f" $R/gpurun_out/${TAG}_kernel_stats.csv
tail -1 /tmp/lr.out
python3 - "$f" "$REPS" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
steps=int(sys.argv[2])+3   # warm-up calls included in the trace
tot=sum(float(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print("all kernels: %d launches, %.2f ms of kernel time over %d steps = %.1f launches and %.2f ms per step" % (calls, tot/1e6, steps, calls/steps, tot/1e6/steps))
for r in rows[:18]: print("%-64s %7s calls %9.1f us total/step %8.1f us avg" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"])/1e3/steps, float(r["AverageNs"])/1e3))
PY
