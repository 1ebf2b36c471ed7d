"""Child process of tests/test_gpu_circom_pipeline.py::test_prove_many_with_items_cut_in_two (GKR_PROVE_MANY_PIECES is read once
per process): the demo circuit's 12 sub-circuits for 40 inputs through gkr_prove_many with the costliest items cut in two, every
proof against the committed digests of the CPU checker's proofs (tests/golden/proof_digests.json, configs[3]'s first 40 inputs)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from gkr_amd import Context, synth
    from gkr_amd.aggregate import ProvingStep
    from gkr_amd.field import as_limbs
    from gkr_amd.prover import _decode_proofs
    n = 40
    golden = synth.proof_digests()["config3"]["digests"]
    step = ProvingStep(synth.mimc7_demo_r1cs())
    inputs = step.inputs_for(np.stack([as_limbs(synth.mimc7_demo_witness(a, b)) for a, b in synth.demo_proof_inputs(64)[:n]]))
    bad = 0
    with Context(0) as ctx:
        for _ in range(2):
            step.prove_raw_many(ctx, inputs, 8)
        for j, (arrs, circuit) in enumerate(zip(step._prepared["outs"], step.circuits)):
            for i, pr in enumerate(_decode_proofs(arrs, circuit.get_k_list())):
                bad += synth.proof_digest(pr.sumcheck_proofs, pr.sumcheck_r, pr.q, pr.z, pr.r)[:16] != golden[i][j]
    step.close()
    print("OK" if bad == 0 else "MISMATCH in %d proofs" % bad)
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
