"""The plain-C dense oracle (oracle/c) against Python big-int arithmetic, the
pure-Python oracles and the golden fixtures.  CPU only."""

import random

import numpy as np
import pytest

from oracle import cdense, dense
from oracle.field import P
from oracle import mimc7
from helpers import ints, layers_of, right_aligned_equal


def test_field_mul_against_bigints():
    rng = random.Random(7)
    edge = [0, 1, 2, P - 1, P - 2, (P + 1) // 2, 1 << 253, (1 << 253) - 1]
    pairs = [(a, b) for a in edge for b in edge] + [(rng.randrange(P), rng.randrange(P)) for _ in range(300)]
    for a, b in pairs:
        assert cdense.fr_mul(a, b) == a * b % P


def test_mimc_constants_and_hash_match_python():
    for i in (0, 1, 2, 45, 90):
        assert cdense.mimc7_constant(i) == mimc7.CTS[i]
    rng = random.Random(8)
    assert cdense.multi_hash([]) == 0
    assert cdense.multi_hash([12, 45, 78, 41]) == 0x284BC1F34F335933A23A433B6FF3EE179D682CD5E5E2FCDD2D964AFA85104BEB
    for n in (1, 2, 3):
        xs = [rng.randrange(P) for _ in range(n)]
        assert cdense.multi_hash(xs) == mimc7.multi_hash(xs)


def test_fill_table_definition():
    t = cdense.fill_table(1000, 0xC0FFEE)
    vals = cdense.from_limbs(t)
    assert all(v < (1 << 253) for v in vals) and len(set(vals)) == 1000

    def mix(z):
        m = (1 << 64) - 1
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
        return z ^ (z >> 31)
    for i in (0, 1, 999):
        limbs = [mix((0xC0FFEE + (4 * i + j + 1) * 0x9E3779B97F4A7C15) & ((1 << 64) - 1)) for j in range(4)]
        limbs[3] &= (1 << 61) - 1
        assert vals[i] == sum(l << (64 * j) for j, l in enumerate(limbs))


@pytest.mark.parametrize("seed", range(4))
def test_c_layer_sumcheck_equals_python_dense(seed):
    rng = random.Random(300 + seed)
    for it in range(6):
        k_i, k = rng.randint(0, 4), rng.randint(1, 3)
        g = 1 << k_i
        gt = [rng.randint(0, 1) for _ in range(g)]
        l = [rng.randrange(1 << k) for _ in range(g)]
        r = [rng.randrange(1 << k) for _ in range(g)]
        z = [rng.randrange(P) for _ in range(k_i)]
        if it % 3 == 0:
            w = [rng.randrange(P) for _ in range(1 << k)]
        elif it % 3 == 1:
            w = [(i >> (k - 1)) + 1 for i in range(1 << k)]
        else:
            w = [rng.randrange(2) for _ in range(1 << k)]
        for threads in (1, 3):
            assert cdense.sumcheck_layer(k_i, k, gt, l, r, z, w, threads) == dense.sumcheck_layer(k_i, k, gt, l, r, z, w)
        A, M = cdense.predicate_tables(k_i, k, gt, l, r, z)
        a, m = dense.predicate_tables(k_i, k, gt, l, r, z)
        assert cdense.from_limbs(A) == a and cdense.from_limbs(M) == m
        b = [rng.randrange(P) for _ in range(k)]
        c = [rng.randrange(P) for _ in range(k)]
        assert cdense.line_restriction(b, c, w, k) == dense.line_restriction(b, c, w, k)


def test_c_mle_sumcheck_equals_python_dense_and_golden(mle_cases):
    rng = random.Random(11)
    for n in (2, 3, 6, 10):
        t = [rng.randrange(P) for _ in range(1 << n)]
        assert cdense.sumcheck_mle(t, n, 2) == dense.sumcheck_mle(t, n)
    for t, n in (([5] * 16, 4), ([i >> 1 for i in range(32)], 5), ([0] * 8, 3)):
        assert cdense.sumcheck_mle(t, n) == dense.sumcheck_mle(t, n)
    for case in mle_cases:
        proof, r = cdense.sumcheck_mle(ints(case["table"]), case["n"])
        assert r == ints(case["r"])
        assert all(right_aligned_equal(a, b) for a, b in zip(proof, ints(case["proof"])))


def test_c_prove_matches_reference_python_fixtures(gkr_cases):
    for case in gkr_cases:
        vals = ints(case["values"])
        if not all(all(dense.depends_on(v, k)) for v, k in zip(vals[1:], case["k"][1:])):
            continue
        out = cdense.prove(layers_of(case), ints(case["inputs"]), z0=ints(case["z0"]))
        assert out["values"] == vals
        assert out["sumcheck_r"] == ints(case["sumcheck_r"])
        assert out["z"] == ints(case["z"]) and out["r"] == ints(case["r"])
        for lay in range(len(out["q"])):
            assert right_aligned_equal(out["q"][lay], ints(case["q"][lay]))
            for mine, ref in zip(out["sumcheck_proofs"][lay], ints(case["sumcheck_proofs"][lay])):
                assert right_aligned_equal(mine, ref)


def test_c_mle_2pow16_threads_agree():
    n = 16
    t = cdense.fill_table(1 << n, 0xC0FFEE + 2)
    c1, l1, r1 = cdense.sumcheck_mle_raw(t, n, 1)
    c2, l2, r2 = cdense.sumcheck_mle_raw(t, n, 0)
    assert np.array_equal(c1, c2) and np.array_equal(l1, l2) and np.array_equal(r1, r2)
    # claim check: g_1(0) + g_1(1) equals the table sum
    vals = cdense.from_limbs(t)
    g1 = cdense.from_limbs(c1[0])
    assert (2 * g1[1] + g1[0]) % P == sum(vals) % P


# ---- the linear-time layer prover (ogkr_sumcheck_layer_lin): the checker for layers wider than the dense form reaches ----

@pytest.mark.parametrize("seed", range(4))
def test_c_linear_time_layer_equals_the_dense_forms(seed):
    rng = random.Random(900 + seed)
    for it in range(8):
        k_i, k = rng.randint(0, 6), rng.randint(1, 4)
        g = 1 << k_i
        gt = [rng.randint(0, 1) for _ in range(g)]
        l = [rng.randrange(1 << k) for _ in range(g)]
        r = [rng.randrange(1 << k) for _ in range(g)]
        z = [rng.randrange(P) for _ in range(k_i)]
        if it % 4 == 0:
            w = [rng.randrange(P) for _ in range(1 << k)]
        elif it % 4 == 1:
            w = [(i >> (k - 1)) + 1 for i in range(1 << k)]     # depends on the first variable only: short round vectors
        elif it % 4 == 2:
            w = [rng.randrange(2) for _ in range(1 << k)]
        else:
            w = [7] * (1 << k)                                   # constant
        if it == 5:
            gt = [0] * g
        if it == 6:
            gt = [1] * g
        ref = dense.sumcheck_layer(k_i, k, gt, l, r, z, w)
        for threads in (1, 3):
            assert cdense.sumcheck_layer_lin(k_i, k, gt, l, r, z, w, threads) == ref
        assert cdense.sumcheck_layer(k_i, k, gt, l, r, z, w) == ref


def test_c_linear_time_layer_equals_dense_c_at_the_dense_forms_limit():
    # the widest layers the O(2^{2k}) form finishes in seconds; raw limbs, no Python integers per entry
    for seed, (k_i, k) in enumerate([(14, 10), (9, 11), (16, 8), (12, 12)]):
        rng = np.random.default_rng(4000 + seed)
        g = 1 << k_i
        gt = rng.integers(0, 2, g, dtype=np.uint8)
        l = rng.integers(0, 1 << k, g, dtype=np.uint32)
        r = rng.integers(0, 1 << k, g, dtype=np.uint32)
        z = cdense.fill_table(max(k_i, 1), 77 + seed)[:k_i]
        w = cdense.fill_table(1 << k, 99 + seed)
        a = cdense.sumcheck_layer_raw(k_i, k, gt, l, r, z, w)
        b = cdense.sumcheck_layer_lin_raw(k_i, k, gt, l, r, z, w)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_c_linear_time_rejects_bad_gates():
    with pytest.raises(ValueError):
        cdense.sumcheck_layer_lin(2, 2, [0, 1, 0, 2], [0, 1, 2, 3], [0, 1, 2, 3], [1, 2], [1, 2, 3, 4])
    with pytest.raises(ValueError):
        cdense.sumcheck_layer_lin(2, 2, [0, 1, 0, 1], [0, 1, 2, 4], [0, 1, 2, 3], [1, 2], [1, 2, 3, 4])


def test_c_prove_raw_equals_prove_and_fixtures(gkr_cases):
    # the limb-only prover built on the linear-time layers = the dense-layer composition = the reference's fixtures
    rng = random.Random(55)
    done = 0
    for case in gkr_cases:
        if ints(case["z0"]) and any(ints(case["z0"])):
            continue   # prove_raw fixes z[0] = 0 (prover.rs:16-21)
        vals = ints(case["values"])
        if not all(all(dense.depends_on(v, k)) for v, k in zip(vals[1:], case["k"][1:])):
            continue
        out = cdense.prove_raw(layers_of(case), cdense.to_limbs(ints(case["inputs"])))
        assert [cdense.from_limbs(v) for v in out["values"]] == vals
        for lay in range(len(out["C"])):
            assert cdense.from_limbs(out["R"][lay]) == ints(case["sumcheck_r"])[lay]
            q = cdense.from_limbs(out["q"][lay])
            assert right_aligned_equal(q[len(q) - out["q_len"][lay]:], ints(case["q"][lay]))
        assert cdense.from_limbs(out["r"]) == ints(case["r"])
        done += 1
    # and random circuits against cdense.prove
    for it in range(4):
        ks = [rng.randint(1, 3) for _ in range(rng.randint(2, 4))]
        layers = []
        for i in range(len(ks) - 1):
            g = 1 << ks[i]
            layers.append(([rng.randint(0, 1) for _ in range(g)], [rng.randrange(1 << ks[i + 1]) for _ in range(g)],
                           [rng.randrange(1 << ks[i + 1]) for _ in range(g)]))
        inputs = [rng.randrange(P) for _ in range(1 << ks[-1])]
        a = cdense.prove(layers, inputs)
        b = cdense.prove_raw(layers, cdense.to_limbs(inputs))
        for lay in range(len(layers)):
            assert cdense.from_limbs(b["R"][lay]) == a["sumcheck_r"][lay]
            rows = [cdense.from_limbs(b["C"][lay][j])[3 - int(b["L"][lay][j]):] for j in range(2 * ks[lay + 1])]
            assert rows == a["sumcheck_proofs"][lay]
            q = cdense.from_limbs(b["q"][lay])
            assert q[len(q) - b["q_len"][lay]:] == a["q"][lay]
            assert cdense.from_limbs(b["z"][lay + 1]) == a["z"][lay + 1]
        assert cdense.from_limbs(b["r"]) == a["r"]
