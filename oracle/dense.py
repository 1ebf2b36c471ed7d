"""Dense-table form of the reference sumcheck (oracle; tests only, pure ints).

Same outputs as oracle/termlist.py (tests/test_oracle_equivalence.py checks
it on random small layers) but O(2^v) per sumcheck instead of exponential-
times-terms, so it reaches 2^16 points in seconds.  The C oracle
(oracle/c/) is this file again with 4x64-bit Montgomery arithmetic.

Layout: index = bit string with variable 1 as the MOST significant bit
(rust/src/gkr/poly.rs:507, 117-131), so round j pairs entry i with i + h,
h = len/2.
"""

from .field import P
from .mimc7 import multi_hash


def mobius_msb(vals, k):
    """Evaluation table -> monomial coefficients (what get_multi_ext stores,
    poly.rs:502-536), index bit (k-1-b) <-> variable b+1."""
    c = [v % P for v in vals]
    for b in range(k):
        bit = 1 << (k - 1 - b)
        for i in range(1 << k):
            if i & bit:
                c[i] = (c[i] - c[i ^ bit]) % P
    return c


def monomial_terms(vals, k):
    """get_multi_ext as a set of [coeff, bits..] (order not significant)."""
    c = mobius_msb(vals, k)
    return [[c[m]] + [(m >> (k - 1 - j)) & 1 for j in range(k)] for m in range(1 << k) if c[m]]


def depends_on(vals, k):
    """dep[b] = True iff some non-zero monomial of the MLE contains variable b+1
    (equivalently: the table differs across bit k-1-b somewhere)."""
    dep = []
    for b in range(k):
        bit = 1 << (k - 1 - b)
        dep.append(any(vals[i] % P != vals[i ^ bit] % P for i in range(1 << k) if not i & bit))
    return dep


def eq_table(z):
    """E[g] = prod_i (z_i if bit_i(g) else 1 - z_i), variable 1 = MSB
    (partial_eval_binary_form, poly.rs:43-62, applied to chi_w_for_binary)."""
    e = [1]
    for zi in z:
        e = [x * f % P for x in e for f in ((1 - zi) % P, zi % P)]
    return e


def predicate_tables(k_i, k_next, gate_type, left, right, z):
    """A[(l<<k)|r] += eq(z,g) for add gates, M[..] for mult gates
    (convert.rs:715-767 + prover.rs:24-37)."""
    n = 1 << (2 * k_next)
    a, m = [0] * n, [0] * n
    e = eq_table(z) if k_i else [1]
    for g, ty in enumerate(gate_type):
        idx = (left[g] << k_next) | right[g]
        if ty:
            m[idx] = (m[idx] + e[g]) % P
        else:
            a[idx] = (a[idx] + e[g]) % P
    return a, m


def sumcheck_layer(k_i, k_next, gate_type, left, right, z, w, python_lengths=False):
    """prove_sumcheck_opt (sumcheck.rs:36-156) on dense tables.

    python_lengths=True reproduces the reference's *Python* prover instead,
    which always emits (and hashes) all three coefficients
    (python/poly.py:168-173); used only to replay fixtures in which a layer's W
    does not depend on some variable, where the two reference provers diverge.

    Returns (proof, r) with the reference's vector lengths: round j has
    2 + dep[(j-1) mod k] coefficients (highest first), see get_univariate_coeff
    (poly.rs:388-420): f1/f2 contribute degree 1 only if a stored monomial of W
    carries the variable; leading zeros that arise numerically are kept.
    """
    k = k_next
    if k < 1:
        raise ValueError("k_next == 0: v = 0 underflows in the reference (sumcheck.rs:49)")
    mask = (1 << k) - 1
    n = 1 << (2 * k)
    a, m = predicate_tables(k_i, k, gate_type, left, right, z)
    f1 = [w[i >> k] % P for i in range(n)]
    f2 = [w[i & mask] % P for i in range(n)]
    dep = depends_on(w, k)
    proof, rs = [], []
    for j in range(2 * k):
        h = len(a) // 2
        c0 = c1 = c2 = 0
        for i in range(h):
            a0, da = a[i], (a[i + h] - a[i])
            m0, dm = m[i], (m[i + h] - m[i])
            p0, dp = f1[i], (f1[i + h] - f1[i])
            q0, dq = f2[i], (f2[i + h] - f2[i])
            s0, ds = p0 + q0, dp + dq
            # a(x) s(x) + m(x) p(x) q(x); only one of dp, dq is non-zero so the
            # cubic term vanishes identically
            pq0 = p0 * q0
            pq1 = p0 * dq + dp * q0
            pq2 = dp * dq
            c0 += a0 * s0 + m0 * pq0
            c1 += a0 * ds + da * s0 + m0 * pq1 + dm * pq0
            c2 += da * ds + m0 * pq2 + dm * pq1
        full = [c2 % P, c1 % P, c0 % P]
        length = 3 if python_lengths else 2 + (1 if dep[j % k] else 0)
        g = full[3 - length:]
        proof.append(g)
        r = multi_hash(g, 0)
        rs.append(r)
        a = [(a[i] + r * (a[i + h] - a[i])) % P for i in range(h)]
        m = [(m[i] + r * (m[i + h] - m[i])) % P for i in range(h)]
        f1 = [(f1[i] + r * (f1[i + h] - f1[i])) % P for i in range(h)]
        f2 = [(f2[i] + r * (f2[i + h] - f2[i])) % P for i in range(h)]
    return proof, rs


def sumcheck_mle(table, n):
    """prove_sumcheck (sumcheck.rs:158-214) for a multilinear g given as its
    2^n evaluations.  Rounds 1..n-1: the summed term list is merged by
    add_poly (poly.rs:293-334), which drops a zero linear coefficient, so
    g_j = [c1, c0] if c1 != 0 else [c0].  Last round (sumcheck.rs:206-207): no
    merge, length 2 iff a stored monomial carries x_n, i.e. iff the table
    depends on its last variable."""
    if n < 2:
        raise ValueError("n < 2 is degenerate in the reference")
    t = [x % P for x in table]
    dep_last = any(t[2 * i] != t[2 * i + 1] for i in range(len(t) // 2))
    proof, rs = [], []
    for j in range(n):
        h = len(t) // 2
        c0 = sum(t[:h]) % P
        c1 = (sum(t[h:]) - c0) % P
        if j < n - 1:
            g = [c1, c0] if c1 else [c0]
        else:
            g = [c1, c0] if dep_last else [c0]
        proof.append(g)
        r = multi_hash(g, 0)
        rs.append(r)
        t = [(t[i] + r * (t[i + h] - t[i])) % P for i in range(h)]
    return proof, rs


def line_restriction(b, c, w, k):
    """reduce_multiple_polynomial (poly.rs:469-500): q(t) = W(b + t(c-b)),
    highest first, length 1 + max total degree of a stored monomial."""
    coeffs = mobius_msb(w, k)
    line = [((ci - bi) % P, bi % P) for bi, ci in zip(b, c)]
    res = [0]
    for mono in range(1 << k):
        if not coeffs[mono]:
            continue
        poly = [coeffs[mono]]
        for j in range(k):
            if (mono >> (k - 1 - j)) & 1:
                g, c0 = line[j]
                nxt = [0] * (len(poly) + 1)
                for i, pc in enumerate(poly):
                    nxt[i] = (nxt[i] + pc * g) % P
                    nxt[i + 1] = (nxt[i + 1] + pc * c0) % P
                poly = nxt
        n = max(len(res), len(poly))
        res = [0] * (n - len(res)) + res
        poly = [0] * (n - len(poly)) + poly
        res = [(x + y) % P for x, y in zip(res, poly)]
    return res


def layer_eval(gate_type, left, right, prev):
    """calculate_input's forward step (convert.rs:812-831)."""
    return [prev[l] * prev[r] % P if ty else (prev[l] + prev[r]) % P
            for ty, l, r in zip(gate_type, left, right)]


def prove(layers, input_values, z0=None, python_lengths=False):
    """prover.rs:6-96 on dense tables.  layers[i] = (gate_type, left, right),
    layer 0 = outputs.  Returns a dict with the Proof fields of gkr.rs:7-19
    (d / input_func as monomial term lists, order not significant).
    z0 defaults to the all-zero vector (prover.rs:16-21); a caller-supplied z0
    exists only so the fixtures made with python/gkr.py's random z[0] can be
    replayed."""
    vals = [[v % P for v in input_values]]
    for gt, l, r in reversed(layers):
        vals.append(layer_eval(gt, l, r, vals[-1]))
    vals.reverse()
    ks = []
    for v in vals:
        k = 0
        while (1 << k) < len(v):
            k += 1
        ks.append(k)
    z = [[0] * ks[0]] if z0 is None else [[v % P for v in z0]]
    sps, srs, qs, rstars = [], [], [], []
    for i, (gt, l, r) in enumerate(layers):
        kn = ks[i + 1]
        sp, sr = sumcheck_layer(ks[i], kn, gt, l, r, z[i], vals[i + 1], python_lengths)
        sps.append(sp)
        srs.append(sr)
        b_star, c_star = sr[:kn], sr[kn:]
        qs.append(line_restriction(b_star, c_star, vals[i + 1], kn))
        r_star = multi_hash(sp[-1], 0)
        z.append([(bi + (ci - bi) * r_star) % P for bi, ci in zip(b_star, c_star)])
        rstars.append(r_star)
    return dict(sumcheck_proofs=sps, sumcheck_r=srs, d=monomial_terms(vals[0], ks[0]), q=qs, z=z,
                r=rstars, depth=len(layers) + 1, input_func=monomial_terms(vals[-1], ks[-1]), k=ks,
                values=vals)
