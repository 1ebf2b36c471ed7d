# SQ counters of the segment gate passes (k_seg_pass<uv>, <rows>, k_seg_combine) on the GPU box, one rocprofv3 --pmc pass per
# counter group, per-launch means:  bash tools/pmc_gate_passes.sh > gpurun_out/seg_pass_pmc_counters.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY"; do
  D=/tmp/pmc_$(echo $G | tr ' ' '_' | cut -c1-40)
  rm -rf $D
  rocprofv3 --pmc $G --output-format csv -d $D -- python3 $R/bench.py --mode layer-split --steps 3 --warmup 1 --no-profile > /dev/null 2>&1
  python3 - "$D" <<'P'
import collections, csv, glob, os, sys
per = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "k_seg_pass" in n:
            n = "k_seg_pass<rows>" if "k_seg_pass<true" in n else "k_seg_pass<uv>"
        elif "k_seg_combine" in n:
            n = "k_seg_combine"
        else:
            continue
        per[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in per.items():
    print(n, {k: int(sum(v) / len(v)) for k, v in sorted(c.items())}, "launches", len(next(iter(c.values()))))
P
done
