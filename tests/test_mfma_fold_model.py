"""Integer model of the matrix-core fold pass (gkr_amd/csrc/mfma_fold.h), checked on the CPU.

The device kernel computes T'[i] = sum_b w_b T[b S + i] mod p as an int8 matrix product over signed byte
digits plus one reduction.  This test re-derives every constant the header hard-codes (column bias tables,
the quotient-estimate multiplier) and replays the kernel's integer steps in Python -- with the 32/64-bit
wrap-arounds the device code has -- on random and extreme operands, so that a wrong constant or a bound
that does not hold shows up without a GPU.  (The GPU parity tests then check the real kernel.)
"""

import os
import random
import re

import pytest

from oracle.field import P

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gkr_amd", "csrc", "mfma_fold.h")
M32 = (1 << 32) - 1
M64 = (1 << 64) - 1


def _header_constants():
    src = open(HEADER).read()
    body = src[src.index("kMfmaColumnBias[kMfmaMaxJ][32] = {"):]
    body = body[:body.index("};")]
    rows = re.findall(r"\{([^{}]*)\}", body)
    table = [[int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", r)] for r in rows]
    mu = int(re.search(r"constexpr uint32_t mu = (0x[0-9a-fA-F]+)u;", src).group(1), 16)
    return table, mu


def test_column_bias_tables_are_multiples_of_p():
    table, _ = _header_constants()
    assert len(table) == 5 and all(len(r) == 32 for r in table)
    for j, row in enumerate(table, start=1):
        bias = 1 << (20 + j)
        assert all(bias <= v < bias + 256 for v in row)
        assert sum(v << (8 * m) for m, v in enumerate(row)) % P == 0
        z = (-(bias * ((1 << 256) - 1) // 255)) % P
        assert [v - bias for v in row] == [(z >> (8 * m)) & 255 for m in range(32)]


def test_quotient_multiplier():
    _, mu = _header_constants()
    d = (P >> 224) + 1
    assert mu == (1 << 61) // d and mu < (1 << 32)


def _signed_digits(r):
    """32 signed radix-256 digits of r < p (mfma_plan_block's recode loop)."""
    out, carry = [], 0
    for m in range(32):
        v = ((r >> (8 * m)) & 0xFF) + carry
        carry = (v + 128) >> 8
        out.append(v - 256 if v >= 128 else v)
    assert carry == 0 and sum(d << (8 * m) for m, d in enumerate(out)) == r
    assert all(-128 <= d <= 127 for d in out)
    return out


def _reduce_274(limbs, mu):
    """mf_reduce_274 with the device's word sizes."""
    p_limbs = [(P >> (32 * i)) & M32 for i in range(8)]
    t1 = limbs[7] * mu
    t2 = limbs[8] * mu
    assert t1 <= M64 and t2 <= M64
    q = ((t2 + (t1 >> 32)) & M64) >> 29
    assert q <= M32
    x = sum(l << (32 * i) for i, l in enumerate(limbs))
    assert 0 <= x // P - q <= 1, "quotient estimate off by more than one"
    r, prod, borrow = [], 0, 0
    for i in range(8):
        prod += q * p_limbs[i]
        assert prod <= M64
        s = prod & M32
        prod >>= 32
        d = (limbs[i] - s - borrow) & M64
        r.append(d & M32)
        borrow = d >> 63
    val = sum(v << (32 * i) for i, v in enumerate(r))
    assert val == x - q * P and val < 2 * P
    return val - P if val >= P else val


def _fold_entry(entries, weights, table_row, mu, j):
    """One output entry the way the kernel forms it; entries / weights: 2^j canonical values."""
    nb = 1 << j
    digits = [[_signed_digits(weights[b] * (1 << (8 * k)) % P) for k in range(32)] for b in range(nb)]
    # start values: 128 * column sums (as 2 * (digits x 64)) + bias table
    start = [2 * sum(64 * digits[b][k][m] for b in range(nb) for k in range(32)) + table_row[m] for m in range(32)]
    cols = list(start)
    for b in range(nb):
        for k in range(32):
            a = ((entries[b] >> (8 * k)) & 0xFF) ^ 0x80
            a_signed = a - 256 if a >= 128 else a
            for m in range(32):
                cols[m] += digits[b][k][m] * a_signed
    assert all(0 < c < (1 << 27) for c in cols), "column sums must stay positive and small"
    assert all(-(1 << 31) <= s < (1 << 31) for s in start)
    # join four columns per limb (64-bit), carry through nine limbs
    limbs, run = [], 0
    for limb in range(8):
        v = sum(cols[4 * limb + t] << (8 * t) for t in range(4))
        assert v < (1 << 52)
        run += v
        limbs.append(run & M32)
        run >>= 32
    assert run <= M32
    limbs.append(run)
    assert sum(l << (32 * i) for i, l in enumerate(limbs)) < (1 << 274)
    return _reduce_274(limbs, mu)


@pytest.mark.parametrize("j", [1, 2, 3, 4, 5])
def test_model_matches_field_arithmetic(j):
    table, mu = _header_constants()
    rng = random.Random(77 + j)
    extremes = [0, 1, P - 1, P - 2, int.from_bytes(b"\x80" * 31 + b"\x20", "little"), int.from_bytes(b"\x7f" * 31 + b"\x2f", "little"),
                int.from_bytes(b"\xff" * 31 + b"\x2f", "little"), (1 << 253) - 1]
    nb = 1 << j
    cases = [([P - 1] * nb, [P - 1] * nb), ([0] * nb, [P - 1] * nb), ([extremes[6]] * nb, [extremes[6]] * nb),
             ([extremes[4]] * nb, [extremes[5]] * nb)]
    for _ in range(6 if j < 5 else 3):
        cases.append(([rng.choice(extremes) if rng.random() < 0.3 else rng.randrange(P) for _ in range(nb)],
                      [rng.choice(extremes) if rng.random() < 0.3 else rng.randrange(P) for _ in range(nb)]))
    for entries, weights in cases:
        want = sum(e * w for e, w in zip(entries, weights)) % P
        assert _fold_entry(entries, weights, table[j - 1], mu, j) == want
