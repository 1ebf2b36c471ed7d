# Counters of the wide layers' product passes on the matrix cores (k_prod_cross_mfma, k_prod_fold_mfma; mfma_cross.h, mfma_fold.h) on
# the GPU box: one rocprofv3 --pmc pass per counter group over tools/bench_wide.py at the given shape, per-launch means per kernel,
# with the kernels' average durations from a separate --kernel-trace --stats run:
#   bash tools/pmc_product_passes.sh 20,20 > gpurun_out/product_pass_pmc_20_20.txt
# FETCH_SIZE / WRITE_SIZE: KiB, FETCH_SIZE doubled for the gfx950 wide-read under-count, as MI355X_MICROARCH.md's HBM section
# prescribes (printed raw and converted).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SHAPE=${1:-20,20}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pst
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pst -- python3 $R/tools/bench_wide.py $SHAPE > /tmp/pst.out 2>&1
tail -1 /tmp/pst.out
python3 - "$(find /tmp/pst -name '*kernel_stats.csv' | head -1)" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_prod_" in r["Name"]:
        print("%-60s calls %4s  avg %9.1f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3))
P
for G in "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_I8 SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  D=/tmp/pmcp_$(echo $G | tr ' ' '_' | cut -c1-40)
  rm -rf $D
  rocprofv3 --pmc $G --output-format csv -d $D -- python3 $R/tools/bench_wide.py $SHAPE > /dev/null 2>&1
  python3 - "$D" <<'P'
import collections, csv, glob, os, sys
per = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "k_prod_" not in n:
            continue
        n = n.split("(")[0].replace("void ", "").replace("gkr::", "")
        if "k_prod_cross_mfma" in n or "k_prod_cross<" in n:
            n += " grid %s" % r.get("Grid_Size", r.get("Grid_Size_X", "?"))
        per[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in sorted(per.items()):
    means = {k: sum(v) / len(v) for k, v in sorted(c.items())}
    extra = ""
    if "FETCH_SIZE" in means:
        extra = "  = %.1f MiB fetched per launch (KiB, x 2: the gfx950 wide-read correction)" % (means["FETCH_SIZE"] * 2 / 1024.0)
    if "WRITE_SIZE" in means:
        extra = "  = %.2f MiB written per launch (KiB)" % (means["WRITE_SIZE"] / 1024.0)
    print(n, {k: int(v) for k, v in means.items()}, "launches", len(next(iter(c.values()))), extra)
P
done
