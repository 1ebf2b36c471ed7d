"""When each proving thread of the concurrent step enters and leaves the library (interpreter overhead around the calls).
    python tools/proof_thread_timeline.py [inputs]"""
import ctypes
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
step = ProvingStep(synth.mimc7_demo_r1cs())
inputs = step.inputs_for(np.stack([as_limbs(synth.mimc7_demo_witness(2 + i, 3 + (i % 5))) for i in range(n_inputs)]))
subs = len(step.circuits)
ctxs = [Context(0) for _ in range(12)]
for c in ctxs:
    c.set_host_threads(1)


def run(log):
    order = sorted(range(subs), key=lambda j: -sum(step.circuits[j].get_k_list()))
    lock = threading.Lock()
    busy = ctypes.c_int32(len(ctxs))
    t0 = time.perf_counter()

    def work(ctx, idx):
        t_in = time.perf_counter()
        while True:
            with lock:
                if not order:
                    break
                j = order.pop(0)
            a = time.perf_counter()
            ctx.prove_batch_raw(step.circuits[j], inputs[j])
            log.append((idx, j, (t_in - t0) * 1e3, (a - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
        with lock:
            busy.value -= 1
        N.lib().gkr_host_help_while(ctypes.byref(busy))
    ts = [threading.Thread(target=work, args=(c, i + 1)) for i, c in enumerate(ctxs[1:])]
    for t in ts:
        t.start()
    work(ctxs[0], 0)
    for t in ts:
        t.join()
    return (time.perf_counter() - t0) * 1e3


for _ in range(3):
    run([])
# the same step through gkr_prove_many (the library's own threads)
for threads in (12, 14, 10):
    with Context(0) as many:
        for _ in range(3):
            step.prove_raw_many(many, inputs, threads)
        t = time.perf_counter()
        for _ in range(5):
            step.prove_raw_many(many, inputs, threads)
        print("gkr_prove_many, %d threads: %.2f ms per step" % (threads, (time.perf_counter() - t) / 5 * 1e3))
for rep in range(3):
    log = []
    total = run(log)
    print("step %.2f ms" % total)
    for idx, j, t_in, a, b in sorted(log, key=lambda x: x[3]):
        print("  thread %2d sub-circuit %2d (depth %d): thread running at %.2f, call %.2f -> %.2f ms" % (idx, j, step.circuits[j].depth(), t_in, a, b))
