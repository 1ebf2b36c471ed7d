"""The large R1CS of bench.py's aggregated_proofs.large_r1cs leg (262 144 constraints -> 16 layered circuits with layers of
2^14 .. 2^16 values, one witness) through gkr_prove_many: ms per step over many repetitions, for several thread counts.
usage: python tools/bench_large_r1cs.py [reps] [threads ...]"""
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
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    thread_counts = [int(x) for x in sys.argv[2:]] or [14]
    nrounds = 65536
    step = ProvingStep(synth.mimc7_demo_r1cs(nrounds=nrounds))
    inputs = step.inputs_for(np.stack([as_limbs(synth.mimc7_demo_witness(2, 3, nrounds=nrounds))]))
    for threads in thread_counts:
        with Context(0) as ctx:
            for _ in range(3):
                step.prove_raw_many(ctx, inputs, threads)
            each = []
            for _ in range(reps):
                t = time.perf_counter()
                step.prove_raw_many(ctx, inputs, threads)
                each.append((time.perf_counter() - t) * 1e3)
        each.sort()
        print({"threads": threads, "min_ms": round(each[0], 3), "median_ms": round(statistics.median(each), 3), "p90_ms": round(each[int(0.9 * len(each))], 3),
               "max_ms": round(each[-1], 3)}, flush=True)
    step.close()


if __name__ == "__main__":
    main()
