"""Times gkr_prove on a circuit with wide layers (k list on the command line, default 18,20,20): wall time per proof on a warm
context (circuit cached), with GKR_DEBUG_TIMING=1 the library's own split of it."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_amd import Context, GKRCircuit, Layer, synth  # noqa: E402


def main():
    ks = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "18,20,20").split(",")]
    circuit, _, wit = synth.wide_circuit(ks)
    with Context(0) as ctx:
        t = time.perf_counter()
        arrs = ctx.prove_batch_raw(circuit, wit, all_arrays=True)
        first = time.perf_counter() - t
        each, fresh = [], []
        for _ in range(5):
            t = time.perf_counter()
            ctx.prove_batch_raw(circuit, wit, out=arrs)       # the caller's proof buffers reused (their pages are mapped)
            each.append(round((time.perf_counter() - t) * 1e3, 3))
        for _ in range(3):
            t = time.perf_counter()
            ctx.prove_batch_raw(circuit, wit)                 # fresh output arrays every call (40 MiB of untouched pages at 2^20 inputs)
            fresh.append(round((time.perf_counter() - t) * 1e3, 3))
    print({"k": ks, "first_ms": round(first * 1e3, 2), "warm_ms_buffers_reused": each, "warm_ms_fresh_buffers": fresh})


if __name__ == "__main__":
    main()
