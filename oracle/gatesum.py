"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The layer sumcheck (prove_sumcheck_opt, rust/src/gkr/sumcheck.rs:36-156) in its linear-time form, restated in
plain Python so that the algebra the product's gate-list kernels and its gate-sharded multi-GPU form rely on is
checked against the dense and term-list oracles on the CPU:

    f(b, c) = a(b, c) (W(b) + W(c)) + m(b, c) W(b) W(c),     a = add_i(z, ., .), m = mult_i(z, ., .)
    sum_c f(b, c) = W(b) U(b) + V(b),   U(b) = sum_c [a(b, c) + m(b, c) W(c)],   V(b) = sum_c a(b, c) W(c)

U, V are sums over the layer's gates (sumcheck.rs:50-63 reduces over the same gate list), so any partition of the
gates gives partial tables that add up -- `partial_uv` / `partial_rows` take a gate range.  With b bound to u,
the rounds over c work on the single row a_u(c) = sum_b eq(u, b) a(b, c), m_u(c) (sumcheck.rs:97-124).
"""

from .dense import depends_on, eq_table
from .field import P
from .mimc7 import multi_hash


def partial_uv(k_i, k, gate_type, left, right, z, w, first=0, count=None):
    """U, V summed over gates first .. first + count - 1 (global gate index g decides eq(z, g))."""
    e = eq_table(z) if k_i else [1]
    n = 1 << k
    U, V = [0] * n, [0] * n
    count = len(gate_type) - first if count is None else count
    for g in range(first, first + count):
        l, r = left[g], right[g]
        if gate_type[g]:
            U[l] = (U[l] + e[g] * w[r]) % P
        else:
            U[l] = (U[l] + e[g]) % P
            V[l] = (V[l] + e[g] * w[r]) % P
    return U, V


def partial_rows(k_i, k, gate_type, left, right, z, u, first=0, count=None):
    """a_u, m_u summed over a gate range: a_u[c] = sum over add gates with right operand c of eq(z, g) eq(u, left)."""
    e = eq_table(z) if k_i else [1]
    eu = eq_table(u)
    n = 1 << k
    A, M = [0] * n, [0] * n
    count = len(gate_type) - first if count is None else count
    for g in range(first, first + count):
        t = e[g] * eu[left[g]] % P
        tgt = M if gate_type[g] else A
        tgt[right[g]] = (tgt[right[g]] + t) % P
    return A, M


def _fold(t, r):
    h = len(t) // 2
    return [(t[i] + r * (t[i + h] - t[i])) % P for i in range(h)]


def rounds_b(U, V, w, dep):
    """The k rounds that bind b, on the completed U, V and W: g(x) = sum_i W_i(x) U_i(x) + V_i(x)."""
    U, V, W = list(U), list(V), [x % P for x in w]
    proof, rs = [], []
    for j in range(len(dep)):
        h = len(U) // 2
        c0 = sum(W[i] * U[i] + V[i] for i in range(h)) % P
        g1 = sum(W[i + h] * U[i + h] + V[i + h] for i in range(h)) % P
        c2 = sum((W[i + h] - W[i]) * (U[i + h] - U[i]) for i in range(h)) % P
        full = [c2, (g1 - c0 - c2) % P, c0]
        g = full if dep[j] else full[1:]
        r = multi_hash(g, 0)
        proof.append(g)
        rs.append(r)
        U, V, W = _fold(U, r), _fold(V, r), _fold(W, r)
    return proof, rs, W[0]


def rounds_c(A, M, w, wu, dep):
    """The k rounds that bind c on the row at b = u; wu = W(u)."""
    A, M, W = list(A), list(M), [x % P for x in w]
    proof, rs = [], []
    for j in range(len(dep)):
        h = len(A) // 2
        c0 = g1 = c2 = 0
        for i in range(h):
            s0, s1 = wu + W[i], wu + W[i + h]
            pq0, pq1 = wu * W[i], wu * W[i + h]
            c0 += A[i] * s0 + M[i] * pq0
            g1 += A[i + h] * s1 + M[i + h] * pq1
            c2 += (A[i + h] - A[i]) * (s1 - s0) + (M[i + h] - M[i]) * (pq1 - pq0)
        c0, g1, c2 = c0 % P, g1 % P, c2 % P
        full = [c2, (g1 - c0 - c2) % P, c0]
        g = full if dep[j] else full[1:]
        r = multi_hash(g, 0)
        proof.append(g)
        rs.append(r)
        A, M, W = _fold(A, r), _fold(M, r), _fold(W, r)
    return proof, rs


def sumcheck_layer(k_i, k, gate_type, left, right, z, w, shards=1, reduce_fr=None):
    """The whole sumcheck from `shards` gate ranges; reduce_fr(list of per-shard vectors) -> their sum (default:
    added here).  Equal to dense.sumcheck_layer for every shard count."""
    dep = depends_on(w, k)
    g = len(gate_type)
    cuts = [g * s // shards for s in range(shards + 1)]
    add = reduce_fr or (lambda parts: [sum(col) % P for col in zip(*parts)])
    uv = add([sum(partial_uv(k_i, k, gate_type, left, right, z, w, cuts[s], cuts[s + 1] - cuts[s]), []) for s in range(shards)])
    n = 1 << k
    pb, rb, wu = rounds_b(uv[:n], uv[n:], w, dep)
    rows = add([sum(partial_rows(k_i, k, gate_type, left, right, z, rb, cuts[s], cuts[s + 1] - cuts[s]), []) for s in range(shards)])
    pc, rc = rounds_c(rows[:n], rows[n:], w, wu, dep)
    return pb + pc, rb + rc
