"""Shared helpers for the parity tests (golden decoding, comparisons)."""

from oracle.field import P


def ints(x):
    if isinstance(x, list):
        return [ints(v) for v in x]
    return int(x)


def layers_of(case):
    return [(l["gate_type"], l["left"], l["right"]) for l in case["layers"]]


def right_aligned_equal(short, full):
    """The reference's Python prover always emits the full-degree vector; the
    Rust prover may emit a shorter one whose dropped leading coefficients are
    provably zero.  Equal iff `full` is `short` left-padded with zeros."""
    if len(short) > len(full):
        return False
    pad = len(full) - len(short)
    return all(v % P == 0 for v in full[:pad]) and [v % P for v in full[pad:]] == [v % P for v in short]


def terms_as_set(terms):
    """Monomial term lists are order-free (HashMap order in the reference)."""
    return {(tuple(t[1:]), t[0] % P) for t in terms if t[0] % P}


# ---- the reference's proof -> verifier.circom input rules, restated for the tests (independent of the library)
def expected_circom_meta(p):
    # aggregator.rs:92-146
    m = [p.depth, max(p.k), p.k[0], len(p.d), max(max(len(t) for t in layer) for layer in p.sumcheck_proofs),
         max(len(q) for q in p.q), len(p.input_func), p.k[p.depth - 1]]
    return m + list(p.k)


def expected_circom_input(p, index):
    # aggregator.rs:148-213, then :49-82 and file_utils.rs:20-28
    m = expected_circom_meta(p)
    sp = []
    for layer in p.sumcheck_proofs:
        rows = [[0] * (m[4] - len(t)) + list(t) for t in layer]
        rows += [[0] * m[4] for _ in range(2 * m[1] - len(layer))]
        sp.append(rows)
    sr = [list(r) + [0] * (2 * m[1] - len(r)) for r in p.sumcheck_r]
    q = [[0] * (m[5] - len(v)) + list(v) for v in p.q]
    z = [list(v) + [0] * (m[1] - len(v)) for v in p.z]
    s = lambda x: str(int(x) % P)
    deep = lambda a: [deep(x) for x in a] if isinstance(a, list) else s(a)
    body = {"sumcheckProof": deep(sp), "sumcheckr": deep(sr), "q": deep(q), "D": deep([list(t) for t in p.d]), "z": deep(z),
            "r": deep(list(p.r)), "inputFunc": deep([list(t) for t in p.input_func])}
    return {k + str(index): v for k, v in body.items()}


def canon_circom(d):
    """D / inputFunc are term lists whose order the reference leaves to a HashMap: compare as sets."""
    out = {}
    for k, v in d.items():
        out[k] = sorted(map(tuple, v)) if k.startswith(("D", "inputFunc")) else v
    return out


