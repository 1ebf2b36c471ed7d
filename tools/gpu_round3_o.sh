#!/bin/bash
O=gpurun_out/r03o; mkdir -p $O
for v in "A=1" "GKR_RETIRE_HELPS=1" "A=2" "GKR_RETIRE_HELPS=1"; do echo "$v"; env $v PROBE_REPS=40 PROBE_THREADS=14 python tools/proof_many_probe.py 64 2>/dev/null | tail -1; done | tee $O/retire_helps.txt
