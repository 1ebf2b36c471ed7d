#!/bin/bash
O=gpurun_out/r03e; mkdir -p $O
for i in 1 2 3; do
 echo '"default"' | tee -a $O/lottery_groups2.jsonl; timeout 300 python tools/lottery_probe.py 8 2>/dev/null | tee -a $O/lottery_groups2.jsonl
 echo '"GKR_NO_LATE_STREAM=1"' | tee -a $O/lottery_groups2.jsonl; GKR_NO_LATE_STREAM=1 timeout 300 python tools/lottery_probe.py 8 2>/dev/null | tee -a $O/lottery_groups2.jsonl
 echo '"GKR_HOST_THREADS=2"' | tee -a $O/lottery_groups2.jsonl; GKR_HOST_THREADS=2 timeout 300 python tools/lottery_probe.py 8 2>/dev/null | tee -a $O/lottery_groups2.jsonl
done
