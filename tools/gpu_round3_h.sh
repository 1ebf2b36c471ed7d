#!/bin/bash
O=gpurun_out/r03h; mkdir -p $O
GKR_DEBUG_TIMING=1 timeout 300 python tools/proof_many_timers.py > /dev/null 2> $O/proof_timers2.txt
grep -n "==== step 6 ====" -A400 $O/proof_timers2.txt | grep -B400 "==== step 6 took" | grep "this thread\|took\|depth=" | head -40
