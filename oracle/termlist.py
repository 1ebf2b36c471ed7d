"""Term-list restatement of the reference Rust prover (oracle; tests only).

A polynomial is a list of terms; a term is ``[coeff, e_1, .., e_L]`` (ints mod
P).  Ordinary polynomials store exponents; the wiring predicates add_i/mult_i
use the *binary form* of rust/src/gkr/poly.rs:26-41: ``e_i = 0`` variable
absent, ``1`` factor ``(1 - x_i)``, ``2`` factor ``x_i``.

Every function names the reference lines it follows.  The code is a
restatement (plain ints, no clones, no HashMap order dependence), small sizes
only: cost is exponential exactly like the reference's.

Rust-only rules restated here and NOT reproducible with the reference's
Python prover (see oracle/__init__.py): ``z[0] = 0`` (prover.rs:16-21) and the
round-vector length rule that falls out of get_univariate_coeff
(poly.rs:388-420).
"""

from .field import P
from .mimc7 import multi_hash

# ----------------------------------------------------------------------------
# poly.rs helpers
# ----------------------------------------------------------------------------


def get_empty(l):
    """poly.rs:12-14 -- one all-zero term of width l+1."""
    return [[0] * (l + 1)]


def chi_w_for_binary(w: str):
    """poly.rs:28-41 -- Lagrange basis term of the bit string w, binary form."""
    return [[1] + [1 if ch == "0" else 2 for ch in w]]


def _bin_factor(code, x):
    if code == 1:
        return (1 - x) % P
    if code == 2:
        return x % P
    return 1


def partial_eval_binary_form(f, x):
    """poly.rs:43-62 -- substitute x for the first len(x) variables; drop them."""
    l = len(x)
    out = []
    for t in f:
        c = t[0]
        for i in range(l):
            c = c * _bin_factor(t[i + 1], x[i]) % P
        out.append([c] + list(t[l + 1:]))
    return out


def partial_eval_i_binary_form(f, x, i):
    """poly.rs:64-83 -- substitute variable i, keep the width, zero its slot."""
    out = []
    for t in f:
        nt = list(t)
        nt[0] = t[0] * _bin_factor(t[i], x) % P
        nt[i] = 0
        out.append(nt)
    return out


def partial_eval_i(f, x, i):
    """poly.rs:160-179."""
    out = []
    for t in f:
        nt = list(t)
        nt[0] = t[0] * pow(x, t[i], P) % P
        nt[i] = 0
        out.append(nt)
    return out


def partial_eval_from(f, r, idx):
    """poly.rs:181-208 -- substitute r for variables idx .. idx+len(r)-1."""
    assert len(f[0]) > len(r)
    if not r:
        return [list(t) for t in f]
    out = []
    for t in f:
        nt = list(t)
        c = t[0]
        for i, ri in enumerate(r):
            e = t[idx + i]
            if e == 0:
                continue
            c = c * pow(ri, e, P) % P
            nt[idx + i] = 0
        nt[0] = c
        out.append(nt)
    return out


def partial_eval_from_binary_form(f, x, idx):
    """poly.rs:210-233."""
    out = []
    for t in f:
        nt = list(t)
        c = t[0]
        for i, xi in enumerate(x):
            code = t[idx + i]
            if code in (1, 2):
                c = c * _bin_factor(code, xi) % P
                nt[idx + i] = 0
        nt[0] = c
        out.append(nt)
    return out


def partial_eval(f, r):
    """poly.rs:235-258 -- substitute the first len(r) variables; drop them."""
    assert len(f[0]) > len(r)
    if not r:
        return [list(t) for t in f]
    out = []
    for t in f:
        c = t[0]
        for i, ri in enumerate(r):
            c = c * pow(ri, t[i + 1], P) % P
        out.append([c] + list(t[len(r) + 1:]))
    return out


def eval_univariate(f, x):
    """poly.rs:260-267 -- Horner, highest degree first."""
    res = f[0]
    for c in f[1:]:
        res = (res * x + c) % P
    return res


def modify_poly_from_k(f, k):
    """poly.rs:269-280 -- shift exponents right by k ("W as a function of c")."""
    return [[t[0]] + [0] * k + list(t[1:]) for t in f]


def extend_length(t, l):
    """poly.rs:282-291."""
    return list(t) + [0] * (l - len(t))


def add_poly(f1, f2):
    """poly.rs:293-334 -- merge by exponent vector, drop zero coefficients.

    The reference iterates a HashMap, so term order is unspecified; here it is
    first-seen order.  Nothing downstream may depend on the order.
    """
    len1 = len(f1[0]) if f1 else 0
    len2 = len(f2[0]) if f2 else 0
    width = max(len1, len2)
    acc = {}
    for t in list(f1) + list(f2):
        te = extend_length(t, width)
        key = tuple(te[1:])
        acc[key] = (acc.get(key, 0) + te[0]) % P
    return [[c] + list(k) for k, c in acc.items() if c != 0]


def get_univariate_coeff(f, i, is_binary_form):
    """poly.rs:388-420 -- coefficient vector in variable i, highest first."""
    if is_binary_form:
        lo, hi = 0, 0          # constant and linear coefficient
        for t in f:
            if t[i] == 1:      # c * (1 - x)
                lo = (lo + t[0]) % P
                hi = (hi - t[0]) % P
            elif t[i] == 2:    # c * x
                hi = (hi + t[0]) % P
        return [hi, lo]
    coeffs = [0]
    for t in f:
        deg = t[i]
        if len(coeffs) - 1 < deg:
            coeffs += [0] * (deg - len(coeffs) + 1)
        coeffs[deg] = (coeffs[deg] + t[0]) % P
    return coeffs[::-1]


def mult_univariate(p, q):
    """poly.rs:422-442 -- schoolbook, highest first."""
    res = [0] * (len(p) + len(q) - 1)
    for i, a in enumerate(p):
        for j, b in enumerate(q):
            res[i + j] = (res[i + j] + a * b) % P
    return res


def add_univariate(p, q):
    """poly.rs:444-467 -- right-aligned sum; the empty vector is the identity."""
    if not p:
        return list(q)
    if not q:
        return list(p)
    n = max(len(p), len(q))
    pp = [0] * (n - len(p)) + list(p)
    qq = [0] * (n - len(q)) + list(q)
    # where only one operand reaches, the reference copies it (same value)
    return [(a + b) % P for a, b in zip(pp, qq)]


def reduce_multiple_polynomial(b, c, w):
    """poly.rs:469-500 -- q(t) = W(b + t (c - b)) as coefficients, highest first."""
    assert len(b) == len(c)
    line = [((ci - bi) % P, bi % P) for bi, ci in zip(b, c)]   # (gradient, const)
    res = [0]
    for term in w:
        poly = [term[0] % P]
        for idx, deg in enumerate(term[1:]):
            for _ in range(deg):
                poly = mult_univariate(poly, [line[idx][0], line[idx][1]])
        res = add_univariate(res, poly)
    return res


def l_function(b, c, r):
    """poly.rs:538-551."""
    return [(bi + (ci - bi) * r) % P for bi, ci in zip(b, c)]


def get_multi_ext(values, v):
    """poly.rs:502-536 (+ chi_w 85-115, mult_mono 336-347).

    Evaluation table (index = bit string read MSB-first, poly.rs:507) ->
    monomial-coefficient term list, zero coefficients dropped.  The reference
    expands each chi_w into 2^{#zeros} signed monomials and merges them in a
    HashMap; the sum is the same.
    """
    acc = {}
    for idx in range(1 << v):
        val = values[idx] % P
        if val == 0:
            continue
        bits = [(idx >> (v - 1 - j)) & 1 for j in range(v)]
        zeros = [j for j in range(v) if bits[j] == 0]
        base = tuple(bits)
        # prod_{j in zeros} (1 - x_j) * prod_{ones} x_j
        for mask in range(1 << len(zeros)):
            e = list(base)
            sign = 1
            for t, j in enumerate(zeros):
                if (mask >> t) & 1:
                    e[j] = 1
                    sign = -sign
            key = tuple(e)
            acc[key] = (acc.get(key, 0) + sign * val) % P
    return [[c] + list(k) for k, c in acc.items() if c != 0]


# ----------------------------------------------------------------------------
# sumcheck.rs
# ----------------------------------------------------------------------------


def n_trailing_bits(wire, n):
    """sumcheck.rs:24-33 -- unique suffixes of length n, first-seen order."""
    seen = []
    have = set()
    for w in wire:
        s = tuple(w[len(w) - n:]) if n else tuple()
        if s not in have:
            have.add(s)
            seen.append(list(s))
    return seen


def _round_sum(wire, pred, f1, f2, nbits, var, is_mult):
    """One map-reduce of sumcheck.rs:50-63 / 65-78 / 97-124."""
    total = []
    for a in n_trailing_bits(wire, nbits):
        f1s = partial_eval_from(f1, a, var + 1)
        f2s = partial_eval_from(f2, a, var + 1)
        ps = partial_eval_from_binary_form(pred, a, var + 1)
        c1 = get_univariate_coeff(f1s, var, False)
        c2 = get_univariate_coeff(f2s, var, False)
        cp = get_univariate_coeff(ps, var, True)
        inner = mult_univariate(c1, c2) if is_mult else add_univariate(c1, c2)
        total = add_univariate(total, mult_univariate(inner, cp))
    return total


def prove_sumcheck_opt(add_wire, mult_wire, add_i, mult_i, f1, f2, v):
    """sumcheck.rs:36-156.  Returns (proof, r)."""
    if v < 1:
        raise ValueError("v == 0 underflows in the reference (sumcheck.rs:49)")
    proof, r = [], []
    g = add_univariate(_round_sum(add_wire, add_i, f1, f2, v - 1, 1, False),
                       _round_sum(mult_wire, mult_i, f1, f2, v - 1, 1, True))
    proof.append(g)
    r.append(multi_hash(g, 0))
    f1j, f2j, aj, mj = f1, f2, add_i, mult_i
    for j in range(1, v - 1):
        f1j = partial_eval_i(f1j, r[-1], len(r))
        f2j = partial_eval_i(f2j, r[-1], len(r))
        aj = partial_eval_i_binary_form(aj, r[-1], len(r))
        mj = partial_eval_i_binary_form(mj, r[-1], len(r))
        g = add_univariate(_round_sum(add_wire, aj, f1j, f2j, v - j - 1, j + 1, False),
                           _round_sum(mult_wire, mj, f1j, f2j, v - j - 1, j + 1, True))
        proof.append(g)
        r.append(multi_hash(g, 0))
    # last round, sumcheck.rs:132-153
    f1v = partial_eval(f1, r)
    f2v = partial_eval(f2, r)
    av = partial_eval_binary_form(add_i, r)
    mv = partial_eval_binary_form(mult_i, r)
    c1 = get_univariate_coeff(f1v, 1, False)
    c2 = get_univariate_coeff(f2v, 1, False)
    ca = get_univariate_coeff(av, 1, True)
    cm = get_univariate_coeff(mv, 1, True)
    g = add_univariate(mult_univariate(add_univariate(c1, c2), ca),
                       mult_univariate(mult_univariate(c1, c2), cm))
    if v == 1:
        # the reference pushes round 1 and then the "last round" again when
        # v == 1; unreachable in practice (v = 2k is even)
        pass
    proof.append(g)
    r.append(multi_hash(g, 0))
    return proof, r


def generate_binary(l):
    """poly.rs:133-158 -- all 0/1 vectors, first variable most significant."""
    return [[(i >> (l - 1 - j)) & 1 for j in range(l)] for i in range(1 << l)] if l else []


def prove_sumcheck(g, v):
    """sumcheck.rs:158-214 (dead code in Rust; python/sumcheck.py:6-53 is its live twin)."""
    proof, r = [], []
    acc = get_empty(v)
    for a in generate_binary(v - 1):
        sub = g
        for i, xi in enumerate(a):
            sub = partial_eval_i(sub, xi, i + 2)
        acc = add_poly(acc, sub)
    # add_poly drops every term when the sum is identically zero; the reference
    # would then index f[..] of an empty list only inside get_univariate_coeff,
    # which handles it (returns [0]).
    cv = get_univariate_coeff(acc, 1, False)
    proof.append(cv)
    r.append(multi_hash(cv, 0))
    for j in range(1, v - 1):
        gj = g
        for i, ri in enumerate(r):
            gj = partial_eval_i(gj, ri, i + 1)
        acc = get_empty(v)
        for a in generate_binary(v - j - 1):
            sub = gj
            for i, xi in enumerate(a):
                sub = partial_eval_i(sub, xi, j + i + 2)
            acc = add_poly(acc, sub)
        cv = get_univariate_coeff(acc, j + 1, False)
        proof.append(cv)
        r.append(multi_hash(cv, 0))
    gv = partial_eval(g, r)
    cv = get_univariate_coeff(gv, 1, False)
    proof.append(cv)
    r.append(multi_hash(cv, 0))
    return proof, r


# ----------------------------------------------------------------------------
# gkr.rs data contract + convert.rs builders + prover.rs
# ----------------------------------------------------------------------------


class Layer:
    """gkr.rs:35-51."""

    def __init__(self, k, add, mult, wire):
        self.k, self.add, self.mult, self.wire = k, add, mult, wire


class GKRCircuit:
    """gkr.rs:53-114."""

    def __init__(self, layer, input_k):
        self.layer, self.input_k = layer, input_k

    def depth(self):
        return len(self.layer)

    def k(self, i):
        return self.input_k if i == len(self.layer) else self.layer[i].k

    def get_k_list(self):
        return [self.k(i) for i in range(self.depth())] + [self.input_k]


class Input:
    """gkr.rs:21-33."""

    def __init__(self, w, d):
        self.w, self.d = w, d


def get_k(n):
    """convert.rs:140-152 -- get_k(1) = 0, else ceil(log2 n)."""
    k = 0
    while (1 << k) < n:
        k += 1
    return k


def build_layer(k_i, k_next, gate_type, left, right):
    """convert.rs:703-777 -- wiring strings, 0/1 wire vectors, add_i / mult_i.

    gate_type[g] is 0 for Add, 1 for Mult; (left[g], right[g]) index layer i+1.
    """
    v = k_i + 2 * k_next
    add_s, mult_s = [], []
    for g, ty in enumerate(gate_type):
        cur = format(g, "0{}b".format(k_i)) if k_i else ""
        s = cur + format(left[g], "0{}b".format(k_next)) + format(right[g], "0{}b".format(k_next))
        (mult_s if ty else add_s).append(s)
    add_wire = [[int(ch) for ch in s] for s in add_s]
    mult_wire = [[int(ch) for ch in s] for s in mult_s]
    add_i = get_empty(v)
    for s in add_s:
        add_i = add_poly(add_i, chi_w_for_binary(s))
    mult_i = get_empty(v)
    for s in mult_s:
        mult_i = add_poly(mult_i, chi_w_for_binary(s))
    if not add_i:
        add_i = get_empty(v)
    if not mult_i:
        mult_i = get_empty(v)
    return Layer(k_i, add_i, mult_i, (add_wire, mult_wire))


def calculate_input(layers, input_values, check_output_zero=True):
    """convert.rs:787-849 -- forward evaluation and per-layer MLE term lists.

    layers[i] = (gate_type, left, right) for layer i (0 = output layer).
    Returns (Input, list of per-layer value vectors with index 0 = outputs).
    """
    vals = [list(v % P for v in input_values)]
    for gate_type, left, right in reversed(layers):
        prev = vals[-1]
        cur = []
        for ty, l, r in zip(gate_type, left, right):
            cur.append(prev[l] * prev[r] % P if ty else (prev[l] + prev[r]) % P)
        vals.append(cur)
    vals.reverse()
    if check_output_zero:
        assert vals[0][0] == 0, "convert.rs:838 asserts d_values[0] == 0"
    w = [get_multi_ext(v, get_k(len(v))) for v in vals]
    return Input(w, w[0]), vals


def build_circuit(layers, n_inputs):
    """layers[i] = (gate_type, left, right); sizes must be powers of two."""
    ks = [get_k(len(l[0])) for l in layers] + [get_k(n_inputs)]
    gl = [build_layer(ks[i], ks[i + 1], *layers[i]) for i in range(len(layers))]
    return GKRCircuit(gl, ks[-1])


class Proof:
    """gkr.rs:7-19."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def prove(circuit, inp, z0=None):
    """prover.rs:6-96.  z0 defaults to zeros (prover.rs:16-21); overridable only
    to replay fixtures made with python/gkr.py's random z[0]."""
    sumcheck_proofs, sumcheck_r, q, r_stars = [], [], [], []
    z = [[0] * circuit.layer[0].k] if z0 is None else [[v % P for v in z0]]
    for i in range(circuit.depth()):
        lay = circuit.layer[i]
        kn = circuit.k(i + 1)
        if len(z[i]) == 0:
            add_res = [list(t) for t in lay.add]
            mult_res = [list(t) for t in lay.mult]
        else:
            add_res = partial_eval_binary_form(lay.add, z[i])
            mult_res = partial_eval_binary_form(lay.mult, z[i])
        w_next = inp.w[i + 1]
        wb = [extend_length(t, 2 * kn + 1) for t in w_next]
        wc = modify_poly_from_k(w_next, kn)
        if not wb:
            wb = [[0] * (2 * kn + 1)]
        if not wc:
            wc = [[0] * (2 * kn + 1)]
        sp, r = prove_sumcheck_opt(lay.wire[0], lay.wire[1], add_res, mult_res, wb, wc, 2 * kn)
        sumcheck_proofs.append(sp)
        sumcheck_r.append(r)
        b_star, c_star = r[:kn], r[kn:]
        q.append(reduce_multiple_polynomial(b_star, c_star, w_next))
        r_star = multi_hash(sp[-1], 0)
        z.append(l_function(b_star, c_star, r_star))
        r_stars.append(r_star)
    return Proof(sumcheck_proofs=sumcheck_proofs, sumcheck_r=sumcheck_r, d=inp.d, q=q, z=z,
                 r=r_stars, depth=circuit.depth() + 1, input_func=inp.w[circuit.depth()],
                 k=circuit.get_k_list())
