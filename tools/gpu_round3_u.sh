#!/bin/bash
python - <<'PY'
import time, numpy as np, sys
sys.path.insert(0,'.')
import torch; torch.cuda.init()
from gkr_amd import Context, synth
from oracle import cdense
for k_i,k in ((26,12),(25,13),(27,12)):
    t=time.time(); lay,z,W=synth.config5_layer(k_i,k); print('gen',round(time.time()-t,1),flush=True)
    with Context(0) as ctx:
        t=time.time(); got=ctx.sumcheck_layer_raw(lay,k,z,W); print(k_i,k,'gpu',round(time.time()-t,2),flush=True)
    t=time.time(); want=cdense.sumcheck_layer_raw(k_i,k,lay.gate_type,lay.left,lay.right,z,W); print('oracle',round(time.time()-t,1),flush=True)
    print(k_i,k,all(np.array_equal(a,b) for a,b in zip(got,want)),flush=True)
PY
