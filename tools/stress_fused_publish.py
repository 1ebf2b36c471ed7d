"""Batched proofs of circuits on both sides of the width thresholds, 400 times each, every repeat compared byte for byte with the
first: the product passes that publish from their last block (arrival counters, one release per block) must never lose a
partial.   python tools/stress_fused_publish.py   (on the GPU box)"""
import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gkr_amd import Context, GKRCircuit, Layer, synth
for ks in ([12, 13, 14], [13, 15, 15], [5, 6, 7, 7], [10, 11, 12]):
    rng = np.random.default_rng(sum(ks))
    layers = []
    for i in range(len(ks) - 1):
        g, m = 1 << ks[i], 1 << ks[i + 1]
        layers.append((rng.integers(0, 2, g, dtype=np.uint8), rng.integers(0, m, g, dtype=np.uint32), rng.integers(0, m, g, dtype=np.uint32)))
    circuit = GKRCircuit([Layer(ks[i], *layers[i]) for i in range(len(ks) - 1)], ks[-1])
    wit = np.stack([synth.rand_fr(np.random.default_rng(7 + b), 1 << ks[-1]) for b in range(3)])
    with Context(0) as ctx:
        ref = ctx.prove_batch_raw(circuit, wit, all_arrays=True)
        bad = 0
        for rep in range(400):
            got = ctx.prove_batch_raw(circuit, wit, all_arrays=True)
            bad += not all(np.array_equal(a, b) for a, b in zip(ref, got))
    print("k =", ks, "mismatching repeats of 400:", bad)
