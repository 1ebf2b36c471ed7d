"""Child process of test_gpu_parity.test_layer_path_variants_match_oracle: the layer path's schedule knobs are
read once per process.  Runs layer sumchecks and a batch of proofs against the C oracle."""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_amd import Context, GKRCircuit, Layer  # noqa: E402
from oracle import cdense  # noqa: E402
from oracle.field import P  # noqa: E402


def main():
    with Context(0) as ctx:
        for seed in range(12):
            rng = random.Random(7000 + seed)
            # (two wider layers: passes that span several blocks and leave 1, 2 or 3 rounds for the last pass)
            k_i, k = {0: (14, 7), 10: (12, 9), 11: (13, 10)}.get(seed) or (rng.randint(0, 9), rng.randint(1, 6))
            g = 1 << k_i
            gt = [rng.randint(0, 1) for _ in range(g)]
            if seed % 4 == 1:
                gt = [0] * g
            if seed % 4 == 2:
                gt = [1] * g
            lay = Layer(k_i, gt, [rng.randrange(1 << k) for _ in range(g)], [rng.randrange(1 << k) for _ in range(g)])
            z = [rng.randrange(P) for _ in range(k_i)]
            w = ([rng.randrange(P) for _ in range(1 << k)], [(i >> (k - 1)) + 1 for i in range(1 << k)],
                 [rng.randrange(2) for _ in range(1 << k)])[seed % 3]
            if ctx.prove_sumcheck_opt(lay, k, z, w) != cdense.sumcheck_layer(k_i, k, gt, lay.left, lay.right, z, w):
                print("MISMATCH layer", seed, k_i, k)
                return 1
        # a batch of proofs of one circuit (batched layer sumchecks, worker pool)
        rng = random.Random(7100)
        ks = [3, 4, 5, 4]
        layers = []
        for i in range(3):
            g, m = 1 << ks[i], 1 << ks[i + 1]
            layers.append(([rng.randint(0, 1) for _ in range(g)], [rng.randrange(m) for _ in range(g)], [rng.randrange(m) for _ in range(g)]))
        circuit = GKRCircuit([Layer(ks[i], *layers[i]) for i in range(3)], ks[-1])
        witnesses = [[rng.randrange(P) for _ in range(1 << ks[-1])] for _ in range(20)]
        for pr, wit in zip(ctx.prove_batch(circuit, witnesses), witnesses):
            ref = cdense.prove(layers, wit)
            if not (pr.sumcheck_proofs == ref["sumcheck_proofs"] and pr.sumcheck_r == ref["sumcheck_r"] and pr.q == ref["q"]
                    and pr.z == ref["z"] and pr.r == ref["r"]):
                print("MISMATCH proof")
                return 1
    print("OK")
    return 0


if __name__ == "__main__":
    sys.exit(main())
