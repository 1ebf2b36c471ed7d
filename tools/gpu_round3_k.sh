#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03k2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --mode layer-split --steps 3 --warmup 1 > /dev/null 2>&1
  f=$(ls /tmp/pmc_$tag/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then python3 - "$f" <<'PY' >> $O/seg_pass_counters.txt
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n=r['Kernel_Name']
    if 'k_seg_pass' in n or 'k_seg_combine' in n:
        key='k_seg_pass<%s>'%('rows' if 'ILb1' in n or '<true' in n else 'uv') if 'k_seg_pass' in n else 'k_seg_combine'
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()}, 'launches', len(next(iter(v.values()))))
PY
  else echo "no csv for $set" >> $O/seg_pass_counters.txt; fi
done
cat $O/seg_pass_counters.txt
