#!/bin/bash
O=gpurun_out/r03o; mkdir -p $O
PROBE_REPS=40 PROBE_THREADS=14,15,16,13 python tools/proof_many_probe.py 64 2>/dev/null | tee $O/probe_threads.txt
GKR_HASH_CHUNK=8 PROBE_REPS=40 PROBE_THREADS=14 python tools/proof_many_probe.py 64 2>/dev/null | tee $O/probe_chunk8.txt
GKR_NO_HELP=1 PROBE_REPS=40 PROBE_THREADS=14 python tools/proof_many_probe.py 64 2>/dev/null | tee $O/probe_nohelp.txt
