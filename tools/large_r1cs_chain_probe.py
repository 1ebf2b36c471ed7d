"""Where the 262 144-constraint step's critical path is: every distinct sub-circuit shape of the large R1CS proven ALONE
(gkr_prove_batch, one witness, warm) -- the floor a lockstep group of that shape cannot go below -- and then, with
GKR_DEBUG_TIMING=1 in the environment, the library's own per-layer timers for one deep circuit alone.
usage: [GKR_DEBUG_TIMING=1] python tools/large_r1cs_chain_probe.py [reps]"""
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_amd import Context, synth  # noqa: E402
from gkr_amd.aggregate import ProvingStep  # noqa: E402
from gkr_amd.field import as_limbs  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    nrounds = 65536
    step = ProvingStep(synth.mimc7_demo_r1cs(nrounds=nrounds))
    inputs = step.inputs_for(np.stack([as_limbs(synth.mimc7_demo_witness(2, 3, nrounds=nrounds))]))
    shapes = {}
    for j, c in enumerate(step.circuits):
        shapes.setdefault(tuple(c.get_k_list()), []).append(j)
    with Context(0) as ctx:
        for ks, members in shapes.items():
            j = members[0]
            c, x = step.circuits[j], inputs[j]
            arrs = ctx.prove_batch_raw(c, x, all_arrays=True)
            for _ in range(2):
                ctx.prove_batch_raw(c, x, out=arrs)
            each = []
            for _ in range(reps):
                t = time.perf_counter()
                ctx.prove_batch_raw(c, x, out=arrs)
                each.append((time.perf_counter() - t) * 1e3)
            print({"k": list(ks), "circuits_of_this_shape": len(members), "alone_ms_median": round(statistics.median(each), 3),
                   "alone_ms_min": round(min(each), 3), "rounds": 2 * sum(ks[1:])}, flush=True)
        if os.environ.get("GKR_DEBUG_TIMING"):
            deep = max(shapes, key=lambda ks: sum(ks[1:]))
            j = shapes[deep][0]
            sys.stderr.write("==== one deep circuit alone, k = %s ====\n" % list(deep))
            sys.stderr.flush()
            ctx.prove_batch_raw(step.circuits[j], inputs[j])
    step.close()


if __name__ == "__main__":
    main()


def subsets():
    """gkr_prove_many on subsets of the step's items: every shape's group on its own, then all together."""
    nrounds = 65536
    step = ProvingStep(synth.mimc7_demo_r1cs(nrounds=nrounds))
    inputs = step.inputs_for(np.stack([as_limbs(synth.mimc7_demo_witness(2, 3, nrounds=nrounds))]))
    shapes = {}
    for j, c in enumerate(step.circuits):
        shapes.setdefault(tuple(c.get_k_list()), []).append(j)
    with Context(0) as ctx:
        sets = [("all", list(range(len(step.circuits))))] + [("only k=%s" % list(ks), js) for ks, js in shapes.items()]
        for threads in (14, 8):
            for name, js in sets:
                work = [(step.circuits[j], inputs[j]) for j in js]
                prepared = ctx.prepare_many(work)
                for _ in range(3):
                    ctx.prove_many_raw(prepared, threads)
                each = []
                for _ in range(20):
                    t = time.perf_counter()
                    ctx.prove_many_raw(prepared, threads)
                    each.append((time.perf_counter() - t) * 1e3)
                print({"items": name, "n": len(js), "threads": threads, "median_ms": round(statistics.median(each), 3), "min_ms": round(min(each), 3)}, flush=True)
    step.close()


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "subsets":
    subsets()
