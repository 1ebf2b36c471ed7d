"""Experiment behind ProvingStep.prove_raw_concurrent's work items: the 64-input proving step with every sub-circuit's
batch cut into `parts` pieces (more, shorter items for the contexts to pick from).
    python tools/proof_split_probe.py [inputs]          (GKR_DEBUG_TIMING=1: the library's per-layer host timers)"""
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_amd import Context, synth  # noqa: E402
from gkr_amd import _native as N  # noqa: E402
from gkr_amd.aggregate import ProvingStep  # noqa: E402
from gkr_amd.field import as_limbs  # noqa: E402

n_inputs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
HELPERS = int(os.environ.get("HELPERS", "0"))   # extra threads that only take pieces of the proving threads' host work
step = ProvingStep(synth.mimc7_demo_r1cs())
inputs = step.inputs_for(np.stack([as_limbs(synth.mimc7_demo_witness(2 + i, 3 + (i % 5))) for i in range(n_inputs)]))
subs = len(step.circuits)


def run(ctxs, parts):
    items = []
    for j in sorted(range(subs), key=lambda j: -sum(step.circuits[j].get_k_list())):
        cut = [n_inputs * p // parts for p in range(parts + 1)]
        items += [(j, cut[p], cut[p + 1]) for p in range(parts) if cut[p + 1] > cut[p]]
    lock = threading.Lock()
    busy = ctypes.c_int32(len(ctxs))

    def work(ctx):
        while True:
            with lock:
                if not items:
                    break
                j, a, b = items.pop(0)
            ctx.prove_batch_raw(step.circuits[j], np.ascontiguousarray(inputs[j][a:b]))
        with lock:
            busy.value -= 1
        if "NO_LEND" not in os.environ:
            N.lib().gkr_host_help_while(ctypes.byref(busy))
    ts = [threading.Thread(target=work, args=(c,)) for c in ctxs[1:]]
    ts += [threading.Thread(target=lambda: N.lib().gkr_host_help_while(ctypes.byref(busy))) for _ in range(HELPERS)]
    for t in ts:
        t.start()
    work(ctxs[0])
    for t in ts:
        t.join()


configs = ((12, 1, 1), (12, 1, 2), (12, 1, 4), (12, [2, 2] + [1] * 10, 1), (12, [2, 2] + [1] * 10, 2), (7, 2, 1), (6, 2, 2), (4, 3, 1), (1, 14, 1))
if os.environ.get("PROBE_CONFIGS"):
    configs = json.loads(os.environ["PROBE_CONFIGS"])
for nctx, threads, parts in configs:
    ctxs = [Context(0) for _ in range(nctx)]
    for i, c in enumerate(ctxs):
        c.set_host_threads(threads[i] if isinstance(threads, list) else threads)
    run(ctxs, parts)
    run(ctxs, parts)
    t = time.perf_counter()
    for _ in range(5):
        run(ctxs, parts)
    dt = (time.perf_counter() - t) / 5
    print(json.dumps({"contexts": nctx, "host_threads_each": threads, "parts": parts, "ms": round(dt * 1e3, 2),
                      "proofs_per_sec": round(n_inputs * subs / dt), "help": "GKR_NO_HELP" not in os.environ, "lend": "NO_LEND" not in os.environ, "helpers": HELPERS}), flush=True)
    for c in ctxs:
        c.close()
