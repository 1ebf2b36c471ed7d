"""Dense-table oracle == term-list restatement of the Rust prover, plus the
verifier relations of python/sumcheck.py:55-70 / sumcheckVerify.circom:17-39 as
size-independent properties.  CPU only."""

import random

import pytest

from oracle import dense, termlist as T
from oracle.field import P
from oracle.mimc7 import multi_hash


def _termlist_layer(k_i, k, gt, l, r, z, w):
    lay = T.build_layer(k_i, k, gt, l, r)
    wt = T.get_multi_ext(w, k)
    add_res = T.partial_eval_binary_form(lay.add, z) if z else lay.add
    mult_res = T.partial_eval_binary_form(lay.mult, z) if z else lay.mult
    wb = [T.extend_length(t, 2 * k + 1) for t in wt] or [[0] * (2 * k + 1)]
    wc = T.modify_poly_from_k(wt, k) or [[0] * (2 * k + 1)]
    return T.prove_sumcheck_opt(lay.wire[0], lay.wire[1], add_res, mult_res, wb, wc, 2 * k)


def _w_of_mode(rng, mode, k):
    n = 1 << k
    if mode == 0:
        return [rng.randrange(P) for _ in range(n)]
    if mode == 1:
        return [rng.randrange(3) for _ in range(n)]
    if mode == 2:
        return [7] * n
    if mode == 3:
        return [0] * n
    return [(i >> (k - 1)) + 1 for i in range(n)]      # depends on x1 only


@pytest.mark.parametrize("seed", range(6))
def test_layer_sumcheck_dense_equals_termlist(seed):
    rng = random.Random(seed)
    for it in range(8):
        k_i, k = rng.randint(0, 3), rng.randint(1, 3)
        g = 1 << k_i
        gt = [rng.randint(0, 1) for _ in range(g)]
        if it % 4 == 0:
            gt = [0] * g
        if it % 4 == 1:
            gt = [1] * g
        l = [rng.randrange(1 << k) for _ in range(g)]
        r = [rng.randrange(1 << k) for _ in range(g)]
        z = [rng.randrange(P) for _ in range(k_i)]
        w = _w_of_mode(rng, rng.randint(0, 4), k)
        assert _termlist_layer(k_i, k, gt, l, r, z, w) == dense.sumcheck_layer(k_i, k, gt, l, r, z, w)


def test_length_rule_edge_cases():
    # W without x1 -> two coefficients per round, zero coefficients kept
    p, _ = dense.sumcheck_layer(1, 1, [0, 1], [0, 1], [1, 1], [7], [3, 3])
    assert [len(v) for v in p] == [2, 2] and p[0] == [99, P - 36]
    p, _ = dense.sumcheck_layer(1, 1, [0, 1], [0, 1], [1, 1], [7], [0, 0])
    assert p == [[0, 0], [0, 0]]
    p, _ = dense.sumcheck_layer(1, 1, [0, 1], [0, 1], [1, 1], [7], [5, 7])
    assert [len(v) for v in p] == [3, 3]
    gates = ([0, 1, 0, 1], [0, 1, 2, 3], [3, 2, 1, 0])
    p, _ = dense.sumcheck_layer(2, 2, *gates, [11, 13], [1, 2, 1, 2])      # 1 + x2
    assert [len(v) for v in p] == [2, 3, 2, 3]
    p, _ = dense.sumcheck_layer(2, 2, *gates, [11, 13], [1, 1, 2, 2])      # 1 + x1
    assert [len(v) for v in p] == [3, 2, 3, 2]
    for w in ([7, 7, 7, 7], [0, 0, 0, 0]):
        p, _ = dense.sumcheck_layer(2, 2, *gates, [11, 13], w)
        assert [len(v) for v in p] == [2, 2, 2, 2]
    with pytest.raises(ValueError):
        dense.sumcheck_layer(1, 0, [0, 1], [0, 0], [0, 0], [3], [5])


@pytest.mark.parametrize("seed", range(4))
def test_mle_sumcheck_dense_equals_termlist(seed):
    rng = random.Random(100 + seed)
    for it in range(6):
        n = rng.randint(2, 5)
        mode = it % 4
        if mode == 0:
            t = [rng.randrange(P) for _ in range(1 << n)]
        elif mode == 1:
            t = [rng.randrange(2) for _ in range(1 << n)]
        elif mode == 2:
            t = [5] * (1 << n)
        else:
            t = [i >> 1 for i in range(1 << n)]            # independent of x_n
        g = T.get_multi_ext(t, n) or T.get_empty(n)
        assert T.prove_sumcheck(g, n) == dense.sumcheck_mle(t, n)


def test_mle_known_vector():
    proof, r = dense.sumcheck_mle(list(range(1, 9)), 3)
    assert proof[0] == [16, 10] and proof[1][0] == 4 and proof[2][0] == 1
    assert r[0] == 12724281212385160142290442537468884417922542462181086529482401478608990951218


def _horner(c, x):
    acc = 0
    for v in c:
        acc = (acc * x + v) % P
    return acc


@pytest.mark.parametrize("seed", range(3))
def test_sumcheck_verifier_relations(seed):
    """g_j(0) + g_j(1) == g_{j-1}(r_{j-1}); r_j == MiMC(g_j); last claim == f(r)."""
    rng = random.Random(200 + seed)
    k_i, k = 2, 3
    g = 1 << k_i
    gt = [rng.randint(0, 1) for _ in range(g)]
    l = [rng.randrange(1 << k) for _ in range(g)]
    r_ = [rng.randrange(1 << k) for _ in range(g)]
    z = [rng.randrange(P) for _ in range(k_i)]
    w = [rng.randrange(P) for _ in range(1 << k)]
    proof, rs = dense.sumcheck_layer(k_i, k, gt, l, r_, z, w)
    a, m = dense.predicate_tables(k_i, k, gt, l, r_, z)
    mask = (1 << k) - 1
    claim = sum(a[i] * (w[i >> k] + w[i & mask]) + m[i] * w[i >> k] * w[i & mask] for i in range(1 << 2 * k)) % P
    for gj, rj in zip(proof, rs):
        assert (_horner(gj, 0) + _horner(gj, 1)) % P == claim
        assert multi_hash(gj, 0) == rj
        claim = _horner(gj, rj)
    # final claim equals add(z,r)(W(b*)+W(c*)) + mult(z,r) W(b*) W(c*)
    eq_b = dense.eq_table(rs[:k])
    eq_c = dense.eq_table(rs[k:])
    wb = sum(e * x for e, x in zip(eq_b, w)) % P
    wc = sum(e * x for e, x in zip(eq_c, w)) % P
    av = sum(a[i] * eq_b[i >> k] * eq_c[i & mask] for i in range(1 << 2 * k)) % P
    mv = sum(m[i] * eq_b[i >> k] * eq_c[i & mask] for i in range(1 << 2 * k)) % P
    assert claim == (av * (wb + wc) + mv * wb * wc) % P
    q = dense.line_restriction(rs[:k], rs[k:], w, k)
    assert _horner(q, 0) == wb and _horner(q, 1) == wc


@pytest.mark.parametrize("seed", range(8))
def test_linear_time_gate_sum_form_equals_dense_for_any_gate_partition(seed):
    """oracle/gatesum.py: the (W, U, V) / single-row form of the layer sumcheck the product's gate-list kernels and
    its gate-sharded multi-GPU form compute -- same transcript as the dense and term-list forms, however the gates
    are partitioned."""
    from oracle import gatesum
    rng = random.Random(300 + seed)
    k_i, k = rng.randint(0, 6), rng.randint(1, 4)
    g = 1 << k_i
    gt = [rng.randint(0, 1) for _ in range(g)]
    l = [rng.randrange(1 << k) for _ in range(g)]
    r = [rng.randrange(1 << k) for _ in range(g)]
    z = [rng.randrange(P) for _ in range(k_i)]
    w = _w_of_mode(rng, seed % 5, k)
    ref = dense.sumcheck_layer(k_i, k, gt, l, r, z, w)
    for shards in (1, 2, 3, 8):
        assert gatesum.sumcheck_layer(k_i, k, gt, l, r, z, w, shards) == ref
    if k_i <= 3 and k <= 3:
        assert _termlist_layer(k_i, k, gt, l, r, z, w) == ref
