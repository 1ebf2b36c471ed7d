# Counters of the wide layers' gate passes (k_items_pass<uv>, <rows> and their combine steps; kernels_wide.hip) on the GPU box: one rocprofv3 --pmc pass
# per counter group over tools/bench_wide.py at the given shape, per-launch means, with the kernel's average duration from a
# separate --kernel-trace --stats run:   bash tools/pmc_wide_gate_passes.sh 24,18 > gpurun_out/wide_gate_pass_pmc_24_18.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SHAPE=${1:-24,18}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/wst
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wst -- python3 $R/tools/bench_wide.py $SHAPE > /tmp/wst.out 2>&1
tail -1 /tmp/wst.out
python3 - "$(find /tmp/wst -name '*kernel_stats.csv' | head -1)" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_items_" in r["Name"] or "k_eq_table" in r["Name"] or "k_prod_cross" in r["Name"]:
        print("%-60s calls %4s  avg %9.1f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3))
P
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  D=/tmp/pmcw_$(echo $G | tr ' ' '_' | cut -c1-40)
  rm -rf $D
  rocprofv3 --pmc $G --output-format csv -d $D -- python3 $R/tools/bench_wide.py $SHAPE > /dev/null 2>&1
  python3 - "$D" <<'P'
import collections, csv, glob, os, sys
per = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "k_items_pass" in n:
            n = "k_items_pass<rows>" if "k_items_pass<true" in n else "k_items_pass<uv>"
        elif "k_items_combine" in n:
            n = "k_items_combine*"
        elif "k_prod_cross<32" in n:
            n = "k_prod_cross<32>"
        else:
            continue
        per[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in sorted(per.items()):
    print(n, {k: int(sum(v) / len(v)) for k, v in sorted(c.items())}, "launches", len(next(iter(c.values()))))
P
done
