"""A whole proof of a circuit with wide layers (bench.py's wide_prove leg on its own): ms per proof, five repetitions.
    python tools/bench_wide_prove.py [18,20,20]"""
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from gkr_amd import Context, synth  # noqa: E402

ks = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "18,20,20").split(","))
circuit, _, wit = synth.wide_circuit(ks)
with Context(0) as ctx:
    arrs = ctx.prove_batch_raw(circuit, wit, all_arrays=True)
    each = []
    for _ in range(5):
        t = time.perf_counter()
        ctx.prove_batch_raw(circuit, wit, out=arrs)
        each.append(round((time.perf_counter() - t) * 1e3, 3))
print({"k": ks, "ms_each": each})
