"""The reference's own argument types at `prove` (gkr_layer_from_wires, gkr_values_from_terms, gkr_terms_from_coeffs) and the
library's C++ verifier (gkr_verify), all host-only: against the oracle's restatement of convert.rs:715-767 / poly.rs:502-536
and against the reference Python prover's own term lists in the golden fixtures (tests/golden/gkr_circuits.json: `D`,
`input_func`, `values` come from /root/reference/python run in the build container)."""
import copy
import random

import numpy as np
import pytest

from gkr_amd import GKRCircuit, GkrError, Layer, Proof, verify
from gkr_amd import _native as N
from gkr_amd.dropin import layer_from_wires, terms_from_coeffs, values_from_terms, verify_native
from gkr_amd.field import to_limbs
from helpers import ints, layers_of, terms_as_set
from oracle import dense, termlist
from oracle.field import P


def test_wire_vectors_give_back_the_gate_arrays(gkr_cases):
    """termlist.build_layer (convert.rs:703-777) makes Layer.wire from (type, left, right); the library decodes it again."""
    rng = random.Random(11)
    cases = [(c["k"], layers_of(c)) for c in gkr_cases]
    for ks in ([0, 1], [3, 2, 4], [5, 5], [4, 1]):
        cases.append((ks, [([rng.randint(0, 1) for _ in range(1 << ks[i])], [rng.randrange(1 << ks[i + 1]) for _ in range(1 << ks[i])],
                            [rng.randrange(1 << ks[i + 1]) for _ in range(1 << ks[i])]) for i in range(len(ks) - 1)]))
    for ks, layers in cases:
        for i, (gt, l, r) in enumerate(layers):
            lay = termlist.build_layer(ks[i], ks[i + 1], gt, l, r)
            add_wire, mult_wire = lay.wire
            rng.shuffle(add_wire)                      # the order of the lists carries nothing
            got = layer_from_wires(ks[i], ks[i + 1], add_wire, mult_wire)
            assert list(got.gate_type) == list(gt) and list(got.left) == list(l) and list(got.right) == list(r), (ks, i)


def test_wire_vectors_that_do_not_describe_a_layer_are_refused():
    lay = termlist.build_layer(2, 2, [0, 1, 1, 0], [0, 1, 2, 3], [3, 2, 1, 0])
    add_wire, mult_wire = lay.wire
    for bad_add, bad_mult in ((add_wire[:1], mult_wire),                                  # a gate missing
                              (add_wire + [add_wire[0]], mult_wire[:1]),                  # a gate named twice
                              ([[2] + add_wire[0][1:]] + add_wire[1:], mult_wire)):      # an entry that is not a bit
        with pytest.raises(GkrError) as e:
            layer_from_wires(2, 2, bad_add, bad_mult)
        assert e.value.status == N.GKR_ERR_INVALID


def test_term_lists_of_the_reference_prover_evaluate_to_its_layer_values(gkr_cases):
    """`input_func` and `D` of the fixtures are get_multi_ext outputs of the REFERENCE's Python (python/poly.py:308-349);
    `values` are its layer values.  (Python rows are [coeff, e_1 .. e_k] as in Rust.)"""
    for case in gkr_cases:
        ks = case["k"]
        vals = ints(case["values"])
        assert values_from_terms(ints(case["input_func"]), ks[-1]) == [v % P for v in vals[-1]], case["name"]
        if ks[0]:      # (k_0 = 0: the reference's Python get_multi_ext of a one-value layer returns the constant 0 whatever the value)
            assert values_from_terms(ints(case["D"]), ks[0]) == [v % P for v in vals[0]], case["name"]


def test_values_from_terms_inverts_get_multi_ext_and_terms_from_coeffs_matches_it():
    rng = random.Random(5)
    for k in (0, 1, 3, 6):
        vals = [rng.randrange(P) if rng.random() < 0.7 else 0 for _ in range(1 << k)]
        terms = termlist.get_multi_ext(vals, k)
        rng.shuffle(terms)
        assert values_from_terms(terms, k) == vals
        assert values_from_terms(terms + terms, k) == [2 * v % P for v in vals]     # equal monomials add up (add_poly)
        # the monomial-coefficient table the prover returns (index bit k-1-j <-> variable j+1) -> the same term set
        table = [0] * (1 << k)
        for t in terms:
            m = 0
            for e in t[1:]:
                m = (m << 1) | e
            table[m] = t[0]
        assert terms_as_set(terms_from_coeffs(to_limbs(table), k)) == terms_as_set(terms)
    assert values_from_terms([], 2) == [0, 0, 0, 0]
    with pytest.raises(GkrError):
        values_from_terms([[1, 2, 0]], 2)                # x_1^2: not multilinear


def _proof_from_oracle(layers, inputs):
    out = dense.prove(layers, inputs)
    ks = out["k"]
    circ = GKRCircuit([Layer(ks[i], *layers[i]) for i in range(len(layers))], ks[-1])
    pr = Proof(sumcheck_proofs=out["sumcheck_proofs"], sumcheck_r=out["sumcheck_r"], d=out["d"], q=out["q"], z=out["z"],
               r=out["r"], depth=out["depth"], input_func=out["input_func"], k=ks)
    return circ, pr


def test_native_verifier_accepts_the_fixture_circuits_proofs(gkr_cases):
    for case in gkr_cases:
        circ, pr = _proof_from_oracle(layers_of(case), ints(case["inputs"]))
        for threads in (1, 3):
            assert verify_native(circ, pr, threads=threads) == (True, 0, 0), case["name"]
        assert verify(pr, circ)


def test_native_verifier_rejects_what_the_python_verifier_rejects(gkr_cases):
    case = next(c for c in gkr_cases if c["name"].startswith("random_k222"))
    circ, pr = _proof_from_oracle(layers_of(case), ints(case["inputs"]))
    assert verify_native(circ, pr)[0]
    expected = {"sumcheck_proofs": 4, "sumcheck_r": 5, "q": 6, "r": 7, "z": 8, "input_func": 9, "d": 4}
    for field, idx in (("sumcheck_proofs", (0, 1, 2)), ("sumcheck_r", (1, 0)), ("q", (0, 0)), ("r", (1,)), ("z", (1, 0)),
                       ("input_func", (0, 0)), ("d", (0, 0))):
        bad = copy.deepcopy(pr)
        tgt = getattr(bad, field)
        for i in idx[:-1]:
            tgt = tgt[i]
        tgt[idx[-1]] = (tgt[idx[-1]] + 1) % P
        ok, layer, check = verify_native(circ, bad)
        assert not ok and check == expected[field], (field, layer, check)
        assert not verify(bad, circ)
    wrong = GKRCircuit([Layer(l.k, [1 - t for t in l.gate_type], l.left, l.right) for l in circ.layer], circ.input_k)
    assert verify_native(wrong, pr)[0] is False and not verify(pr, wrong)
    # z[0] must be the zero vector (prover.rs:16-21)
    bad = copy.deepcopy(pr)
    bad.z[0][0] = 1
    assert verify_native(circ, bad) == (False, 0, 3)


def test_native_verifier_on_a_wider_random_circuit_with_many_threads():
    rng = np.random.default_rng(3)
    ks = [6, 9, 8]
    layers = [(rng.integers(0, 2, 1 << ks[i]).tolist(), rng.integers(0, 1 << ks[i + 1], 1 << ks[i]).tolist(),
               rng.integers(0, 1 << ks[i + 1], 1 << ks[i]).tolist()) for i in range(2)]
    inputs = [int(x) for x in rng.integers(1, 1 << 62, 1 << ks[-1])]
    from oracle import cdense
    out = cdense.prove(layers, inputs)
    circ = GKRCircuit([Layer(ks[i], *layers[i]) for i in range(2)], ks[-1])
    # (cdense.prove returns the layers' values, not d / input_func: the oracle's get_multi_ext of the outputs and inputs)
    pr = Proof(sumcheck_proofs=out["sumcheck_proofs"], sumcheck_r=out["sumcheck_r"], d=termlist.get_multi_ext(out["values"][0], ks[0]),
               q=out["q"], z=out["z"], r=out["r"], depth=3, input_func=termlist.get_multi_ext(out["values"][-1], ks[-1]), k=ks)
    assert verify_native(circ, pr, threads=8) == (True, 0, 0)
    pr.sumcheck_proofs[1][7][0] = (pr.sumcheck_proofs[1][7][0] + 1) % P
    assert verify_native(circ, pr, threads=8) == (False, 1, 4)
