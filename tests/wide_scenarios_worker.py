"""Child process of tests/test_gpu_wide_layers.py::test_lane_group_passes_on_small_layers, run with
GKR_GATE_GROUPS_MIN_K=1: every layer takes the lane-group gate passes (kernels_wide.hip) whatever its width."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_amd import Context, GKRCircuit, Layer, synth  # noqa: E402
from oracle import cdense  # noqa: E402


def same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


def small():
    ok = True
    with Context(0) as ctx:
        for seed, (k_i, k) in enumerate([(4, 3), (0, 3), (7, 4), (10, 6), (16, 8), (12, 12), (9, 11), (3, 1)]):
            lay, z, W = synth.config5_layer(k_i, k, seed=300 + seed)
            want = cdense.sumcheck_layer_raw(k_i, k, lay.gate_type, lay.left, lay.right, z, W)
            ok &= same(ctx.sumcheck_layer_raw(lay, k, z, W), want)
            ok &= same(cdense.sumcheck_layer_lin_raw(k_i, k, lay.gate_type, lay.left, lay.right, z, W), want)
    return ok


def heavy_batch():
    # skewed buckets (one left operand with a quarter of the gates, one right operand with another quarter) in a batch of
    # proofs: units, their partial sums and the combine step per proof
    rng = np.random.default_rng(4243)
    ks = [17, 9, 7]
    layers = []
    for i in range(2):
        g, m = 1 << ks[i], 1 << ks[i + 1]
        layers.append([rng.integers(0, 2, g, dtype=np.uint8), rng.integers(0, m, g, dtype=np.uint32), rng.integers(0, m, g, dtype=np.uint32)])
    layers[0][1][: 1 << 15] = 5
    layers[0][2][1 << 15: 1 << 16] = 300
    layers = [tuple(x) for x in layers]
    circuit = GKRCircuit([Layer(ks[i], *layers[i]) for i in range(2)], ks[-1])
    wit = np.stack([synth.rand_fr(np.random.default_rng(900 + b), 1 << ks[-1]) for b in range(3)])
    with Context(0) as ctx:
        sc, sl, sr, q, ql, z, rr, dco, ico = ctx.prove_batch_raw(circuit, wit, all_arrays=True)
    ok = True
    for b in range(3):
        ref = cdense.prove_raw(layers, wit[b])
        ro = 0
        for i in range(2):
            k = ks[i + 1]
            ok &= np.array_equal(sc[b, ro:ro + 2 * k], ref["C"][i]) and np.array_equal(sr[b, ro:ro + 2 * k], ref["R"][i])
            ok &= np.array_equal(sl[b, ro:ro + 2 * k], ref["L"][i])
            ro += 2 * k
        ok &= np.array_equal(rr[b], ref["r"])
    return bool(ok)


def prove_wide():
    # whole proofs through layers of 2^13 .. 2^15 values (an even and an odd number of variables bound by the line
    # restriction's launches over the grid): sumchecks, q and its length, r against the C checker
    ok = True
    for seed, ks in enumerate(([12, 13, 14], [13, 15, 15])):
        rng = np.random.default_rng(5150 + seed)
        layers = []
        for i in range(2):
            g, m = 1 << ks[i], 1 << ks[i + 1]
            layers.append((rng.integers(0, 2, g, dtype=np.uint8), rng.integers(0, m, g, dtype=np.uint32), rng.integers(0, m, g, dtype=np.uint32)))
        circuit = GKRCircuit([Layer(ks[i], *layers[i]) for i in range(2)], ks[-1])
        wit = np.stack([synth.rand_fr(np.random.default_rng(77 + seed), 1 << ks[-1])])
        with Context(0) as ctx:
            sc, sl, sr, q, ql, z, rr, dco, ico = ctx.prove_batch_raw(circuit, wit, all_arrays=True)
        ref = cdense.prove_raw(layers, wit[0])
        ro = qo = 0
        for i in range(2):
            k = ks[i + 1]
            ok &= np.array_equal(sc[0, ro:ro + 2 * k], ref["C"][i]) and np.array_equal(sr[0, ro:ro + 2 * k], ref["R"][i])
            ok &= int(ql[0, i]) == ref["q_len"][i] and np.array_equal(q[0, qo:qo + k + 1], ref["q"][i])
            ro += 2 * k
            qo += k + 1
        ok &= np.array_equal(rr[0], ref["r"])
    return bool(ok)


if __name__ == "__main__":
    ok = {"small": small, "heavy-batch": heavy_batch, "per-round": small, "prove-wide": prove_wide}[sys.argv[1]]()
    print("OK" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)
