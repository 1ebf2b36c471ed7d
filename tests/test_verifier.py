"""The host-side GKR verifier (gkr_amd/verifier.py, mirror of python/gkr.py:202-231) on CPU: it accepts the
reference prover's own proofs (golden fixtures, via the oracle's Rust-shaped restatement) and rejects
tampered ones."""

import copy

from gkr_amd import GKRCircuit, Layer, Proof, verify
from oracle import dense
from oracle.field import P
from helpers import ints, layers_of


def _proof_from_oracle(layers, inputs):
    out = dense.prove(layers, inputs)
    ks = out["k"]
    circ = GKRCircuit([Layer(ks[i], *layers[i]) for i in range(len(layers))], ks[-1])
    pr = Proof(sumcheck_proofs=out["sumcheck_proofs"], sumcheck_r=out["sumcheck_r"], d=out["d"], q=out["q"], z=out["z"],
               r=out["r"], depth=out["depth"], input_func=out["input_func"], k=ks)
    return circ, pr


def test_verifier_accepts_oracle_proofs_of_fixture_circuits(gkr_cases):
    n = 0
    for case in gkr_cases:
        circ, pr = _proof_from_oracle(layers_of(case), ints(case["inputs"]))   # z[0] = 0 as in the Rust prover
        assert verify(pr, circ), case["name"]
        n += 1
    assert n >= 10


def test_verifier_rejects_tampering(gkr_cases):
    case = next(c for c in gkr_cases if c["name"].startswith("random_k222"))
    circ, pr = _proof_from_oracle(layers_of(case), ints(case["inputs"]))
    assert verify(pr, circ)
    for field, idx in (("sumcheck_proofs", (0, 1, 2)), ("sumcheck_r", (1, 0)), ("q", (0, 0)), ("r", (1,)), ("z", (1, 0)),
                       ("input_func", (0, 0)), ("d", (0, 0))):
        bad = copy.deepcopy(pr)
        tgt = getattr(bad, field)
        for i in idx[:-1]:
            tgt = tgt[i]
        tgt[idx[-1]] = (tgt[idx[-1]] + 1) % P
        assert not verify(bad, circ), field
    wrong = GKRCircuit([Layer(l.k, [1 - t for t in l.gate_type], l.left, l.right) for l in circ.layer], circ.input_k)
    assert not verify(pr, wrong)
