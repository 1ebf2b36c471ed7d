# Where the segment gate passes' operand gathers are served from (configs[4], k_seg_pass): L2 requests / hits / misses and the
# vector L1's requests to L2, one rocprofv3 --pmc pass per group, per-launch means.
#   bash tools/pmc_seg_pass_memory.sh > gpurun_out/seg_pass_pmc_memory.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for G in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum" "FETCH_SIZE" "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES"; do
  D=/tmp/pmcm_$(echo $G | tr ' ' '_' | cut -c1-40)
  rm -rf $D
  rocprofv3 --pmc $G --output-format csv -d $D -- python3 $R/bench.py --mode layer-split --steps 3 --warmup 1 --no-profile > /dev/null 2>&1
  python3 - "$D" "$G" <<'P'
import collections, csv, glob, os, sys
per = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "k_seg_pass" in n:
            n = "k_seg_pass<rows>" if "k_seg_pass<true" in n else "k_seg_pass<uv>"
        else:
            continue
        per[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
if not per:
    print("(no rows for: %s)" % sys.argv[2])
for n, c in per.items():
    print(n, {k: int(sum(v) / len(v)) for k, v in sorted(c.items())}, "launches", len(next(iter(c.values()))))
P
done
