#!/bin/bash
# rocprofv3 kernel stats of tools/bench_wide.py at the given shapes (default 20,20) -> gpurun_out/wide_stats_<shape>.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for shape in ${@:-20,20}; do
  rm -rf /tmp/ws
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ws -- python3 $R/tools/bench_wide.py $shape > /tmp/ws.out 2>&1
  f=$(find /tmp/ws -name "*kernel_stats.csv" | head -1)
  mkdir -p $R/gpurun_out
  cp "$f" $R/gpurun_out/wide_stats_${shape/,/_}.csv
  tail -1 /tmp/ws.out
  python3 - "$f" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]: print("%-72s %6s %12s %10s" % (r["Name"][:72], r["Calls"], r["TotalDurationNs"], r["AverageNs"]))
PY
done
