"""GKR verifier for the proofs this library produces (host side, plain integers).

Mirrors the reference's Python verifier -- python/gkr.py:202-231 and
python/sumcheck.py:55-70 (the circom one, gkr-verifier-circuits/circom/circom/verifier.circom:39-71,
checks the same relations inside a circuit) -- on the Rust prover's Proof (rust/src/gkr.rs:7-19),
which carries no `f` field: the last sumcheck claim is compared with
add(z, b*, c*) (q(0) + q(1)) + mult(z, b*, c*) q(0) q(1) directly.

Verification is O(gates * k) per layer of cheap arithmetic; it is not on the accelerated path.  It gives an
end-to-end check of a proof that needs neither the CPU checker under tests/ nor the reference.
"""

from typing import List

from .field import MODULUS as P
from .prover import GKRCircuit, Proof, multi_hash


def eval_univariate(coeffs: List[int], x: int) -> int:
    """Horner, highest degree first (rust/src/gkr/poly.rs:260-267)."""
    acc = 0
    for c in coeffs:
        acc = (acc * x + c) % P
    return acc


def eval_expansion(terms: List[List[int]], point: List[int]) -> int:
    """sum_t coeff_t * prod_i point_i^{e_ti} (python/poly.py:293-305)."""
    total = 0
    for t in terms:
        v = t[0] % P
        for e, x in zip(t[1:], point):
            if e:
                v = v * pow(x, e, P) % P
        total = (total + v) % P
    return total


def _eq_bits(point: List[int], index: int) -> int:
    """prod_i (bit_i(index) ? point_i : 1 - point_i), variable 1 = most significant bit."""
    k = len(point)
    v = 1
    for i, x in enumerate(point):
        v = v * (x if (index >> (k - 1 - i)) & 1 else 1 - x) % P
    return v


def wiring_at(layer, z: List[int], b: List[int], c: List[int]):
    """(add_i, mult_i) of a layer evaluated at (z, b, c): sum over its gates of
    eq(z, g) eq(b, left_g) eq(c, right_g)  (chi_w_for_binary terms, rust/src/gkr/poly.rs:28-41)."""
    k = len(b)
    eq_b = [_eq_bits(b, i) for i in range(1 << k)]
    eq_c = [_eq_bits(c, i) for i in range(1 << k)]
    add = mult = 0
    for g, (ty, l, r) in enumerate(zip(layer.gate_type, layer.left, layer.right)):
        w = _eq_bits(z, g) * eq_b[int(l)] % P * eq_c[int(r)] % P
        if int(ty):
            mult = (mult + w) % P
        else:
            add = (add + w) % P
    return add, mult


def verify_sumcheck(claim: int, rounds: List[List[int]], challenges: List[int]):
    """python/sumcheck.py:55-70.  Returns (ok, final claim g_v(r_v))."""
    expected = claim % P
    for g, r in zip(rounds, challenges):
        if (eval_univariate(g, 0) + eval_univariate(g, 1)) % P != expected:
            return False, expected
        if multi_hash(g, 0) != r % P:
            return False, expected
        expected = eval_univariate(g, r)
    return True, expected


def verify(proof: Proof, circuit: GKRCircuit) -> bool:
    """python/gkr.py:202-231 on the Rust-shaped proof."""
    L = circuit.depth()
    if proof.depth != L + 1 or proof.k != circuit.get_k_list():
        return False
    if len(proof.z[0]) != proof.k[0] or any(x % P for x in proof.z[0]):
        return False                       # the Rust prover fixes z[0] = 0 (prover.rs:16-21)
    m = eval_expansion(proof.d, proof.z[0])
    for i in range(L):
        k = proof.k[i + 1]
        rounds, rs = proof.sumcheck_proofs[i], proof.sumcheck_r[i]
        if len(rounds) != 2 * k or len(rs) != 2 * k:
            return False
        ok, last = verify_sumcheck(m, rounds, rs)
        if not ok:
            return False
        b_star, c_star = rs[:k], rs[k:]
        q = proof.q[i]
        q0, q1 = eval_univariate(q, 0), eval_univariate(q, 1)
        add, mult = wiring_at(circuit.layer[i], proof.z[i], b_star, c_star)
        if last != (add * (q0 + q1) + mult * q0 % P * q1) % P:
            return False
        if proof.r[i] % P != multi_hash(rounds[-1], 0):
            return False
        if [x % P for x in proof.z[i + 1]] != [(bi + proof.r[i] * (ci - bi)) % P for bi, ci in zip(b_star, c_star)]:
            return False
        m = eval_univariate(q, proof.r[i])
    return m == eval_expansion(proof.input_func, proof.z[L])
