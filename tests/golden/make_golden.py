#!/usr/bin/env python3
"""Generate tests/golden/*.json by running the REFERENCE'S OWN Python prover.

Run in the build container only (needs /root/reference; nothing under
tests/ reads /root/reference at test time -- the JSON files are the fixtures):

    python tests/golden/make_golden.py

What is executed: /root/reference/python/{poly,sumcheck,gkr}.py, imported
unmodified from where they lie.  Their single missing third-party import,
``ethsnarks`` (``field.FQ`` and ``mimc.mimc_hash``; not installed, not pinned
by the reference), is provided by the two small stand-in modules built below.
The stand-in is a dependency shim, not reference code:

  * ``FQ``: integers mod the BN254 scalar modulus with the operators the
    reference uses (+ - * ** == hash int repr, zero/one/random).
  * ``FQ.random()`` returns the next value of a per-case list (the reference
    draws z[0] at random, python/gkr.py:142-143; the Rust prover fixes
    z[0] = 0, rust/src/gkr/prover.rs:16-21).
  * ``mimc_hash(xs)`` = circomlib-MiMC7 ``multi_hash(xs[1:], key=0)``.  The
    Python prover's round vectors are ``[constant, c2, c1, c0]`` with a leading
    constant slot that is always 0 in this flow (python/poly.py:168-173); the
    Rust prover hashes ``[c2, c1, c0]`` (rust/src/gkr/sumcheck.rs:84).
    Skipping that slot lines the Python transcript up with the Rust one, so
    the same fixture pins both.  The hash itself is the oracle's MiMC7
    (oracle/mimc7.py), pinned separately by public known answers.

Outputs (all integers as decimal strings):
  gkr_circuits.json   toy circuit of python/test_gkr.py + random layered
                      circuits driven through python/gkr.py:prove
  mle_sumcheck.json   python/sumcheck.py:prove_sumcheck on python/poly.py:get_ext
                      of small evaluation tables
"""

import json
import os
import random
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_PY = "/root/reference/python"

sys.path.insert(0, REPO)
from oracle.field import P                      # noqa: E402
from oracle.mimc7 import multi_hash             # noqa: E402

_random_queue = []


class FQ:
    __slots__ = ("n",)

    def __init__(self, n=0):
        self.n = (n.n if isinstance(n, FQ) else int(n)) % P

    @staticmethod
    def _v(o):
        return o.n if isinstance(o, FQ) else int(o)

    def __add__(self, o):
        return FQ(self.n + self._v(o))

    __radd__ = __add__

    def __sub__(self, o):
        return FQ(self.n - self._v(o))

    def __rsub__(self, o):
        return FQ(self._v(o) - self.n)

    def __mul__(self, o):
        if isinstance(o, (FQ, int)):
            return FQ(self.n * self._v(o))
        return NotImplemented

    __rmul__ = __mul__

    def __pow__(self, e):
        return FQ(pow(self.n, self._v(e), P))

    def __neg__(self):
        return FQ(-self.n)

    def __eq__(self, o):
        if isinstance(o, (FQ, int)):
            return self.n == self._v(o) % P
        return NotImplemented

    def __ne__(self, o):
        r = self.__eq__(o)
        return r if r is NotImplemented else not r

    def __hash__(self):
        return hash(self.n)

    def __int__(self):
        return self.n

    def __repr__(self):
        return str(self.n)

    @classmethod
    def zero(cls):
        return cls(0)

    @classmethod
    def one(cls):
        return cls(1)

    @classmethod
    def random(cls):
        return cls(_random_queue.pop(0) if _random_queue else 0)


def _install_standin():
    pkg = types.ModuleType("ethsnarks")
    field = types.ModuleType("ethsnarks.field")
    field.FQ = FQ
    field.SNARK_SCALAR_FIELD = P
    mimc = types.ModuleType("ethsnarks.mimc")
    mimc.mimc_hash = lambda xs, *a, **k: multi_hash([int(x) for x in xs][1:], 0)
    pkg.field, pkg.mimc = field, mimc
    sys.modules["ethsnarks"] = pkg
    sys.modules["ethsnarks.field"] = field
    sys.modules["ethsnarks.mimc"] = mimc


_install_standin()
sys.path.insert(0, REF_PY)
import gkr as ref_gkr            # noqa: E402  (reference module)
import sumcheck as ref_sumcheck  # noqa: E402  (reference module)
import poly as ref_poly          # noqa: E402  (reference module)


def bits_of(i, k):
    return [(i >> (k - 1 - j)) & 1 for j in range(k)]


def table_func(vals, k):
    tbl = {tuple(bits_of(i, k)): FQ(v) for i, v in enumerate(vals)}

    def f(arr):
        return tbl[tuple(int(x) for x in arr)]
    return f


def predicate_func(points):
    s = set(points)

    def f(arr):
        return FQ(1) if tuple(int(x) for x in arr) in s else FQ(0)
    return f


def run_reference_gkr(layers, inputs, z0):
    """layers[i] = (gate_type, left, right), layer 0 = outputs; gate_type 0 add / 1 mult."""
    vals = [[v % P for v in inputs]]
    for gt, l, r in reversed(layers):
        prev = vals[-1]
        vals.append([(prev[a] * prev[b] if t else prev[a] + prev[b]) % P for t, a, b in zip(gt, l, r)])
    vals.reverse()
    ks = [max(0, (len(v) - 1).bit_length()) for v in vals]
    depth = len(vals)
    c = ref_gkr.Circuit(depth)
    for li, v in enumerate(vals):
        for gi, x in enumerate(v):
            c.add_node(li, gi, bits_of(gi, ks[li]), x)
        c.layers[li].add_func(table_func(v, ks[li]))
    for li, (gt, l, r) in enumerate(layers):
        addp, multp = [], []
        for g, (t, a, b) in enumerate(zip(gt, l, r)):
            pt = tuple(bits_of(g, ks[li]) + bits_of(a, ks[li + 1]) + bits_of(b, ks[li + 1]))
            (multp if t else addp).append(pt)
        c.layers[li].add = predicate_func(addp)
        c.layers[li].mult = predicate_func(multp)
    c.layers[depth - 1].add = predicate_func([])
    c.layers[depth - 1].mult = predicate_func([])
    _random_queue[:] = list(z0)
    proof = ref_gkr.prove(c, table_func(vals[0], ks[0]))
    ok = ref_gkr.verify(proof)
    return vals, ks, proof, ok


def S(x):
    return str(int(x))


def dump_case(name, layers, inputs, z0):
    vals, ks, proof, ok = run_reference_gkr(layers, inputs, z0)
    # python/gkr.py:verify evaluates D through get_multi_ext, which returns an
    # all-zero term for a 0-variable output layer (python/poly.py:308-349), so
    # the reference verifier cannot accept k[0] == 0 circuits; the prover side
    # is still recorded for them.
    assert ok or ks[0] == 0, "reference verifier rejected its own proof"
    # every leading constant slot must be zero for the x[1:] adapter to be sound
    for sp in proof.sumcheck_proofs:
        for vec in sp:
            assert int(vec[0]) == 0
    for qv in proof.q:
        assert int(qv[0]) == 0
    return {
        "name": name,
        "layers": [{"gate_type": list(gt), "left": list(l), "right": list(r)} for gt, l, r in layers],
        "inputs": [S(v) for v in inputs],
        "z0": [S(v) for v in z0],
        "k": ks,
        "values": [[S(x) for x in v] for v in vals],
        "reference_verifier_accepts": bool(ok),
        # Python vectors with the leading constant slot removed
        "sumcheck_proofs": [[[S(x) for x in vec[1:]] for vec in sp] for sp in proof.sumcheck_proofs],
        "sumcheck_r": [[S(x) for x in r] for r in proof.sumcheck_r],
        "q": [[S(x) for x in qv[1:]] for qv in proof.q],
        "z": [[S(x) for x in zz] for zz in proof.z],
        "r": [S(x) for x in proof.r],
        "D": [[S(x) for x in t] for t in proof.D],
        "input_func": [[S(x) for x in t] for t in proof.input_func],
    }


def random_layers(rng, ks, mix):
    layers = []
    for i in range(len(ks) - 1):
        g = 1 << ks[i]
        n = 1 << ks[i + 1]
        if mix == "mult":
            gt = [1] * g
        elif mix == "add":
            gt = [0] * g
        else:
            gt = [rng.randint(0, 1) for _ in range(g)]
        layers.append((gt, [rng.randrange(n) for _ in range(g)], [rng.randrange(n) for _ in range(g)]))
    return layers


def main():
    rng = random.Random(0xC0FFEE)
    cases = []
    # the reference's own toy circuit, python/test_gkr.py:7-112
    toy_layers = [([1, 1], [0, 2], [1, 3]), ([1, 1, 1, 1], [0, 1, 1, 3], [0, 1, 2, 3])]
    cases.append(dump_case("test_gkr_toy_z0_zero", toy_layers, [3, 2, 3, 1], [0]))
    cases.append(dump_case("test_gkr_toy_z0_random", toy_layers, [3, 2, 3, 1], [rng.randrange(P)]))
    shapes = [
        ([1, 1], "mix"), ([1, 2], "mix"), ([2, 1], "mix"), ([2, 2], "mult"), ([2, 2], "add"),
        ([0, 1, 2], "mix"), ([1, 2, 2], "mix"), ([2, 2, 2], "mix"), ([1, 1, 1, 1], "mix"),
        ([3, 2], "mix"), ([2, 3], "mix"), ([1, 3, 2], "mix"),
    ]
    for si, (ks, mix) in enumerate(shapes):
        layers = random_layers(rng, ks, mix)
        inputs = [rng.randrange(P) for _ in range(1 << ks[-1])]
        z0 = [0] * ks[0] if si % 2 == 0 else [rng.randrange(P) for _ in range(ks[0])]
        cases.append(dump_case("random_k{}_{}_{}".format("".join(map(str, ks)), mix, si), layers, inputs, z0))
        print("case", cases[-1]["name"], "ok")
    # round 6: wider and deeper circuits through the same reference prover, from a generator of their own (the cases above and
    # the plain-sumcheck fixtures below keep their bytes)
    rng2 = random.Random(0xC0FFEE + 6)
    more = [
        ([3, 3], "mix"), ([2, 3, 3], "mix"), ([3, 3, 3], "mult"), ([1, 2, 3, 2], "mix"), ([3, 2, 3], "add"), ([4, 3], "mix"),
        ([3, 4], "mix"), ([2, 2, 2, 2, 2], "mix"), ([5, 4], "mix"), ([4, 5, 3], "mix"), ([0, 3, 4], "mult"),
    ]
    for si, (ks, mix) in enumerate(more):
        layers = random_layers(rng2, ks, mix)
        # (every third case: a witness with few distinct values -- layers whose extension lacks variables, short q)
        inputs = [rng2.randrange(P) for _ in range(1 << ks[-1])] if si % 3 != 2 else [rng2.randrange(2) for _ in range(1 << ks[-1])]
        z0 = [0] * ks[0] if si % 2 == 0 else [rng2.randrange(P) for _ in range(ks[0])]
        cases.append(dump_case("r6_k{}_{}_{}".format("".join(map(str, ks)), mix, si), layers, inputs, z0))
        print("case", cases[-1]["name"], "ok")
    with open(os.path.join(HERE, "gkr_circuits.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "modulus": S(P), "cases": cases}, f, indent=1)

    mle = []
    tables = [list(range(1, 9))]
    for n in (2, 3, 4, 5, 6):
        tables.append([rng.randrange(P) for _ in range(1 << n)])
    tables.append([rng.randrange(1, 100) for _ in range(16)])
    # round 6: larger tables, from a generator of their own: the seven cases above keep their bytes.  (Tables that do not depend
    # on a variable cannot be pinned here: the Rust prover drops a round vector's leading zero coefficient before hashing it
    # (poly.rs:388-420), this Python prover hashes [0, c0] -- the transcripts part at that round.  Those cases are held against
    # the oracle's restatement of the Rust rule in tests/test_gpu_parity.py.)
    rng3 = random.Random(0xC0FFEE + 60)
    tables.append([rng3.randrange(P) for _ in range(1 << 7)])
    tables.append([rng3.randrange(P) for _ in range(1 << 8)])
    for tbl in tables:
        n = (len(tbl) - 1).bit_length()
        g = ref_poly.get_ext(table_func(tbl, n), n)
        proof, r = ref_sumcheck.prove_sumcheck(g, n, 1)
        for vec in proof:
            assert int(vec[0]) == 0
        claim = sum(tbl) % P
        assert ref_sumcheck.verify_sumcheck(FQ(claim), proof, r, n)
        mle.append({"n": n, "table": [S(x) for x in tbl],
                    "proof": [[S(x) for x in vec[1:]] for vec in proof], "r": [S(x) for x in r]})
    with open(os.path.join(HERE, "mle_sumcheck.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "modulus": S(P), "cases": mle}, f, indent=1)
    print("wrote", len(cases), "gkr cases and", len(mle), "mle cases")


if __name__ == "__main__":
    main()
