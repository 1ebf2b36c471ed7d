"""Child process of tests/test_gpu_config_scale.py::test_segment_passes_scenarios_match_oracle (run with
GKR_GATE_SEGMENTS_MIN_LOG2=16 so that layers of 2^19 / 2^20 gates take the segment form of the gate passes)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_amd import Context, GKRCircuit, Layer, synth  # noqa: E402
from gkr_amd.field import from_limbs  # noqa: E402
from oracle import cdense  # noqa: E402


def skewed():
    k_i, k = 20, 10
    lay, z, W = synth.config5_layer(k_i, k, seed=77)
    gt, l, r = (a.copy() for a in lay.arrays())
    l[: 1 << 18] = 5
    r[1 << 18: 1 << 19] = 1023
    l[l == 7] = 8
    gt[3 << 18:] = 1
    want = cdense.sumcheck_layer_raw(k_i, k, gt, l, r, z, W)
    with Context(0) as ctx:
        got = ctx.sumcheck_layer_raw(Layer(k_i, gt, l, r), k, z, W)
    return all(np.array_equal(a, b) for a, b in zip(got, want))


def batch():
    rng = np.random.default_rng(4242)
    ks = [19, 9, 7]
    layers = []
    for i in range(2):
        g, m = 1 << ks[i], 1 << ks[i + 1]
        layers.append((rng.integers(0, 2, g, dtype=np.uint8), rng.integers(0, m, g, dtype=np.uint32), rng.integers(0, m, g, dtype=np.uint32)))
    circuit = GKRCircuit([Layer(ks[i], *layers[i]) for i in range(2)], ks[-1])
    witnesses = [from_limbs(synth.rand_fr(np.random.default_rng(900 + b), 1 << ks[-1])) for b in range(3)]
    with Context(0) as ctx:
        proofs = ctx.prove_batch(circuit, witnesses)
    plain = [(list(map(int, t)), list(map(int, l)), list(map(int, r))) for t, l, r in layers]
    for pr, w in zip(proofs, witnesses):
        ref = cdense.prove(plain, w)
        if not (pr.sumcheck_proofs == ref["sumcheck_proofs"] and pr.sumcheck_r == ref["sumcheck_r"] and pr.q == ref["q"]
                and pr.z == ref["z"] and pr.r == ref["r"]):
            return False
    return True


if __name__ == "__main__":
    ok = {"skewed": skewed, "batch": batch}[sys.argv[1]]()
    print("OK" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)
