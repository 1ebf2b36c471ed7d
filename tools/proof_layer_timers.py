"""The library's per-layer host timers (GKR_DEBUG_TIMING) for the concurrent proving step: where a round's time goes
when 12 contexts prove the 12 sub-circuits side by side.   GKR_DEBUG_TIMING=1 python tools/proof_layer_timers.py [inputs] [contexts]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_amd import Context, synth  # noqa: E402
from gkr_amd.aggregate import ProvingStep  # noqa: E402
from gkr_amd.field import as_limbs  # noqa: E402

n_inputs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nctx = int(sys.argv[2]) if len(sys.argv) > 2 else 12
step = ProvingStep(synth.mimc7_demo_r1cs())
inputs = step.inputs_for(np.stack([as_limbs(synth.mimc7_demo_witness(2 + i, 3 + (i % 5))) for i in range(n_inputs)]))
ctxs = [Context(0) for _ in range(nctx)]
for c in ctxs:
    c.set_host_threads(1)
for _ in range(3):
    step.prove_raw_concurrent(ctxs, inputs) if nctx > 1 else step.prove_raw(ctxs[0], inputs)
sys.stderr.write("==== measured call ====\n")
sys.stderr.flush()
step.prove_raw_concurrent(ctxs, inputs) if nctx > 1 else step.prove_raw(ctxs[0], inputs)
