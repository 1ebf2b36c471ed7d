"""GKR_DEBUG_TIMING=1 python tools/proof_many_timers.py: gkr_prove_many steps until one is slow; the library's host
timers of that step are the last block on stderr."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_amd import Context, synth  # noqa: E402
from gkr_amd.aggregate import ProvingStep  # noqa: E402
from gkr_amd.field import as_limbs  # noqa: E402

step = ProvingStep(synth.mimc7_demo_r1cs())
inputs = step.inputs_for(np.stack([as_limbs(synth.mimc7_demo_witness(2 + i, 3 + (i % 5))) for i in range(64)]))
with Context(0) as ctx:
    for _ in range(3):
        step.prove_raw_many(ctx, inputs, 12)
    for rep in range(40):
        sys.stderr.write("==== step %d ====\n" % rep)
        sys.stderr.flush()
        t = time.perf_counter()
        step.prove_raw_many(ctx, inputs, 12)
        ms = (time.perf_counter() - t) * 1e3
        sys.stderr.write("==== step %d took %.2f ms ====\n" % (rep, ms))
        if ms > 17.0:
            break
