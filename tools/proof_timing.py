"""Where a batch of small GKR proofs spends its time: bench.py's aggregated-proofs leg under the library's
per-kernel profile (HIP events), printed as time per kernel and the share the kernels are of the wall time."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from gkr_amd import Context, GKRCircuit, Layer  # noqa: E402

n_proofs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ks = [5, 6, 7, 7, 7]
rng = np.random.default_rng(0xC0FFEE + 3)
layers = [Layer(ks[i], rng.integers(0, 2, 1 << ks[i], dtype=np.uint8), rng.integers(0, 1 << ks[i + 1], 1 << ks[i], dtype=np.uint32),
                rng.integers(0, 1 << ks[i + 1], 1 << ks[i], dtype=np.uint32)) for i in range(4)]
circuit = GKRCircuit(layers, ks[-1])
inputs = np.stack([np.random.default_rng(1000 + i).integers(0, 1 << 61, (1 << ks[-1], 4), dtype=np.uint64) for i in range(n_proofs)])
ctx = Context(0)
ctx.prove_batch_raw(circuit, inputs)
reps = 5
t0 = time.perf_counter()
for _ in range(reps):
    ctx.prove_batch_raw(circuit, inputs)
wall = (time.perf_counter() - t0) / reps
print("%d proofs: %.2f ms per batch = %.0f proofs/s, %d sumcheck rounds" % (n_proofs, wall * 1e3, n_proofs / wall, 2 * sum(ks[1:])))
ctx.profile(1)
ctx.prove_batch_raw(circuit, inputs)
ctx.profile_reset()
t0 = time.perf_counter()
ctx.prove_batch_raw(circuit, inputs)
wall_p = time.perf_counter() - t0
tot = 0.0
for name in ("gate_lists", "gate_uv", "gate_rows", "layer_c_round", "layer_round", "layer_round_fused", "layer_uv", "layer_uv_round", "layer_collapse", "layer_fold", "layer_round_reduce",
             "layer_round_hash", "predicate_sorted",
             "predicate_scatter", "predicate_normalise"):
    p = ctx.profile_get(name)
    if p["launches"]:
        tot += p["total_ms"]
        print("  %-22s %4d launches  %8.3f ms  (%.1f us each)" % (name, p["launches"], p["total_ms"], p["total_ms"] * 1e3 / p["launches"]))
print("  kernels %.2f ms of %.2f ms wall (profiled run)" % (tot, wall_p * 1e3))
ctx.close()
