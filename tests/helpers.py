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
