#!/bin/bash
python - <<'PY'
import time, ctypes, numpy as np
from gkr_amd import Context
from gkr_amd import _native as N
with Context(0) as ctx:
    ctx.set_transcript(N.GKR_TRANSCRIPT_DEVICE)
    n=12; count=1<<n
    d=ctx.alloc(count*32); ctx.fill_table(d,count,5); ctx.synchronize()
    ctx.sumcheck_mle_batch_device(d,n,1)
    ts=[]
    for _ in range(5):
        t=time.perf_counter(); ctx.sumcheck_mle_batch_device(d,n,1); ts.append(time.perf_counter()-t)
    print('device transcript, n=12, batch 1: ms per sumcheck', round(min(ts)*1e3,3), 'per round', round(min(ts)*1e3/n,3))
PY
