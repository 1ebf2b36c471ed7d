"""Pin the CPU oracle against fixtures produced by the reference's own Python
prover (tests/golden/make_golden.py; python/gkr.py:130-200,
python/sumcheck.py:6-53).  CPU only."""

import pytest

from oracle import dense, termlist
from oracle.field import P
from helpers import ints, layers_of, right_aligned_equal, terms_as_set


def _check_proof(case, proofs, rs, q, z, rstar):
    g_proofs, g_rs = ints(case["sumcheck_proofs"]), ints(case["sumcheck_r"])
    assert len(proofs) == len(g_proofs)
    for lay in range(len(proofs)):
        assert len(proofs[lay]) == len(g_proofs[lay]) == 2 * case["k"][lay + 1]
        for j, (mine, ref) in enumerate(zip(proofs[lay], g_proofs[lay])):
            assert right_aligned_equal(mine, ref), (case["name"], lay, j)
        assert rs[lay] == g_rs[lay], (case["name"], lay)
        assert right_aligned_equal(q[lay], ints(case["q"][lay])), (case["name"], lay)
    assert z == ints(case["z"])
    assert rstar == ints(case["r"])


def _generic(case):
    """True iff every layer's W depends on all of its variables, i.e. the Rust
    length rule (poly.rs:388-420) and the Python prover's fixed length agree."""
    vals = ints(case["values"])
    return all(all(dense.depends_on(v, k)) for v, k in zip(vals[1:], case["k"][1:]))


def test_fixture_mix(gkr_cases):
    flags = [_generic(c) for c in gkr_cases]
    assert sum(flags) >= 10 and not all(flags)   # both kinds are present


def test_dense_oracle_matches_reference_python(gkr_cases):
    for case in gkr_cases:
        # where W lacks a variable the Rust prover hashes a shorter vector than
        # the Python prover; replay those fixtures with the Python lengths
        out = dense.prove(layers_of(case), ints(case["inputs"]), z0=ints(case["z0"]),
                          python_lengths=not _generic(case))
        assert out["values"] == ints(case["values"])
        _check_proof(case, out["sumcheck_proofs"], out["sumcheck_r"], out["q"], out["z"], out["r"])
        if case["k"][0] > 0:
            assert terms_as_set(out["d"]) == terms_as_set(ints(case["D"]))
        assert terms_as_set(out["input_func"]) == terms_as_set(ints(case["input_func"]))


def test_termlist_oracle_matches_reference_python(gkr_cases):
    for case in gkr_cases:
        if not _generic(case):
            continue    # Rust and Python provers diverge there (see above)
        layers = layers_of(case)
        circuit = termlist.build_circuit(layers, len(case["inputs"]))
        inp, vals = termlist.calculate_input(layers, ints(case["inputs"]), check_output_zero=False)
        assert vals == ints(case["values"])
        pr = termlist.prove(circuit, inp, z0=ints(case["z0"]))
        _check_proof(case, pr.sumcheck_proofs, pr.sumcheck_r, pr.q, pr.z, pr.r)
        assert pr.k == case["k"] and pr.depth == len(case["k"])
        assert terms_as_set(pr.input_func) == terms_as_set(ints(case["input_func"]))


def test_toy_circuit_first_round_is_the_documented_one(gkr_cases):
    toy = next(c for c in gkr_cases if c["name"] == "test_gkr_toy_z0_zero")
    # python/test_gkr.py:7-112: outputs 36, 6; first round of layer 0
    assert ints(toy["values"])[0] == [36, 6]
    assert ints(toy["sumcheck_proofs"])[0][0] == [12, P - 48, 36]
    assert toy["reference_verifier_accepts"] is True


def test_mle_oracles_match_reference_python(mle_cases):
    for case in mle_cases:
        n, table = case["n"], ints(case["table"])
        proof, r = dense.sumcheck_mle(table, n)
        for mine, ref in zip(proof, ints(case["proof"])):
            assert right_aligned_equal(mine, ref)
        assert r == ints(case["r"])
        g = termlist.get_multi_ext(table, n) or termlist.get_empty(n)
        proof2, r2 = termlist.prove_sumcheck(g, n)
        assert proof2 == proof and r2 == r
