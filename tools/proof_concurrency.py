"""aggregated proofs/sec of the demo circuit (64 inputs x 12 sub-circuits) by how many contexts prove sub-circuits
concurrently and how many host threads each may use.   python tools/proof_concurrency.py [inputs]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_amd import Context, synth  # noqa: E402
from gkr_amd.aggregate import ProvingStep  # noqa: E402
from gkr_amd.field import as_limbs  # noqa: E402

os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("GKR_HW_QUEUES", "4"))
n_inputs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
step = ProvingStep(synth.mimc7_demo_r1cs())
inputs = step.inputs_for(np.stack([as_limbs(synth.mimc7_demo_witness(2 + i, 3 + (i % 5))) for i in range(n_inputs)]))
subs = len(step.circuits)
for nctx, threads in ((1, 0), (1, 1), (1, 4), (2, 1), (2, 4), (3, 1), (3, 4), (4, 1), (4, 3), (6, 1), (6, 2), (12, 1)):
    ctxs = [Context(0) for _ in range(nctx)]
    for c in ctxs:
        c.set_host_threads(threads)
    run = (lambda: step.prove_raw(ctxs[0], inputs)) if nctx == 1 else (lambda: step.prove_raw_concurrent(ctxs, inputs))
    run()
    run()
    t = time.perf_counter()
    for _ in range(5):
        run()
    dt = (time.perf_counter() - t) / 5
    print(json.dumps({"contexts": nctx, "host_threads_each": threads or "default", "ms": round(dt * 1e3, 2),
                      "proofs_per_sec": round(n_inputs * subs / dt)}), flush=True)
    for c in ctxs:
        c.close()
