"""CPU-side checks of the product library: it loads, exports every symbol the
header declares, and its host-resident pieces (MiMC7, the device arithmetic run
on the host) agree with big-int arithmetic.  No compute entry point that needs
a GPU is called here."""

import ctypes
import os
import random
import re

import numpy as np
import pytest

import gkr_amd
from gkr_amd import _native as N
from gkr_amd.field import from_limbs, to_limbs
from oracle import mimc7
from oracle.field import P

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(REPO, "include", "gkr_amd.h")).read()
    declared = set(re.findall(r"\b(gkr_[a-z0-9_]+)\s*\(", header))
    declared -= {"gkr_fr", "gkr_ctx"}
    assert declared == set(N.SYMBOLS), declared ^ set(N.SYMBOLS)
    lib = N.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), name


def test_no_device_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(gkr_amd.GkrError) as e:
        gkr_amd.Context(0)
    assert e.value.status == N.GKR_ERR_NO_DEVICE


def test_product_does_not_import_the_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "gkr_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(root, f)).read()
                assert "oracle" not in text.replace("no oracle", ""), os.path.join(root, f)


def test_device_arithmetic_on_host_matches_bigints():
    lib = N.lib()
    rng = random.Random(3)
    edge = [0, 1, 2, P - 1, P - 2, (1 << 253) - 1, 1 << 253, 0xFFFFFFFF, 1 << 32, (1 << 224) - 1]
    pairs = [(a, b) for a in edge for b in edge] + [(rng.randrange(P), rng.randrange(P)) for _ in range(500)]
    for a, b in pairs:
        A, B, O = to_limbs([a]), to_limbs([b]), np.zeros((1, 4), dtype=np.uint64)
        assert lib.gkr_selftest_mul(_p(A), _p(B), _p(O)) == 0
        assert from_limbs(O)[0] == a * b % P
    for n in (0, 1, 1023, 1024, 5000):
        vals = [rng.randrange(P) for _ in range(n)] if n != 1024 else [P - 1] * n
        V = to_limbs(vals) if n else np.zeros((0, 4), dtype=np.uint64)
        O = np.zeros((1, 4), dtype=np.uint64)
        assert lib.gkr_selftest_wide_sum(_p(V), ctypes.c_size_t(n), _p(O)) == 0
        assert from_limbs(O)[0] == sum(vals) % P


def test_fixed_multiplier_fold_on_host_matches_bigints():
    """lo + r (hi - lo) via the round's multiplier table (80-product schedule of the fold kernels)."""
    lib = N.lib()
    rng = random.Random(5)
    edge = [0, 1, P - 1, (1 << 253) + 12345, P - 2]
    cases = [(a, b, r) for a in edge for b in edge for r in edge]
    cases += [(rng.randrange(P), rng.randrange(P), rng.randrange(P)) for _ in range(400)]
    for lo, hi, r in cases:
        L, H, R, O = to_limbs([lo]), to_limbs([hi]), to_limbs([r]), np.zeros((1, 4), dtype=np.uint64)
        assert lib.gkr_selftest_fold(_p(L), _p(H), _p(R), _p(O)) == 0
        assert from_limbs(O)[0] == (lo + r * (hi - lo)) % P


def test_lazy_dot_product_on_host_matches_bigints():
    lib = N.lib()
    rng = random.Random(6)
    for n in (0, 1, 2, 7, 8, 5, 300):
        a = [rng.randrange(P) for _ in range(n)]
        b = [rng.randrange(P) for _ in range(n)]
        if n in (7, 8):
            a, b = [P - 1] * n, [P - 1] * n
        A = to_limbs(a) if n else np.zeros((0, 4), dtype=np.uint64)
        B = to_limbs(b) if n else np.zeros((0, 4), dtype=np.uint64)
        O = np.zeros((1, 4), dtype=np.uint64)
        assert lib.gkr_selftest_dot(_p(A), _p(B), ctypes.c_size_t(n), _p(O)) == 0
        assert from_limbs(O)[0] == sum(x * y for x, y in zip(a, b)) % P


def test_gate_segment_item_on_host_matches_bigints():
    """One item of a sorted gate list's segment (csrc/gate_seg.h) on the host twins of the device code: the selected
    lazy accumulate, the "no second factor" term added at 2^256, the partial reduction with its two top-limb folds (worst
    case: 32 add gates of p - 1), the combine step with E_hi."""
    lib = N.lib()
    rng = random.Random(66)
    for trial in range(40):
        n = [0, 1, 32, 32, 32, 17][trial] if trial < 6 else rng.randint(1, 32)
        e = [rng.randrange(P) for _ in range(n)]
        t = [rng.randrange(P) for _ in range(n)]
        m = [rng.randint(0, 1) for _ in range(n)]
        if trial == 2:
            e, t, m = [P - 1] * n, [P - 1] * n, [0] * n
        if trial == 3:
            e, t, m = [P - 1] * n, [P - 1] * n, [1] * n
        eh = rng.randrange(P) if trial != 4 else P - 1
        E = to_limbs(e) if n else np.zeros((0, 4), dtype=np.uint64)
        T = to_limbs(t) if n else np.zeros((0, 4), dtype=np.uint64)
        M = np.asarray(m, dtype=np.uint8)
        for rows in (0, 1):
            O0, O1 = np.zeros((1, 4), dtype=np.uint64), np.zeros((1, 4), dtype=np.uint64)
            assert lib.gkr_selftest_seg_item(_p(E), _p(T), _p(M), ctypes.c_size_t(n), _p(to_limbs([eh])), ctypes.c_int(rows), _p(O0), _p(O1)) == 0
            if rows:
                want0 = eh * sum(x * y for x, y, mm in zip(e, t, m) if not mm) % P
                want1 = eh * sum(x * y for x, y, mm in zip(e, t, m) if mm) % P
            else:
                want0 = eh * sum(x * (y if mm else 1) for x, y, mm in zip(e, t, m)) % P
                want1 = eh * sum(x * y for x, y, mm in zip(e, t, m) if not mm) % P
            assert from_limbs(O0)[0] == want0 and from_limbs(O1)[0] == want1, (trial, rows)


def test_batched_transcript_hash_matches_oracle():
    """the eight-lane (AVX-512 IFMA where available) MiMC7 of the host transcript, every length mix"""
    lib = N.lib()
    rng = random.Random(8)
    for trial in range(12):
        lens = [rng.randint(0, 3) for _ in range(8)]
        vals = [[rng.randrange(P) for _ in range(3)] for _ in range(8)]
        if trial == 0:
            vals[0] = [0, 0, 0]
            vals[1] = [P - 1, P - 1, P - 1]
            lens[0], lens[1] = 3, 3
        V = to_limbs([x for row in vals for x in row])
        L = np.asarray(lens, dtype=np.uint32)
        O = np.zeros((8, 4), dtype=np.uint64)
        used = ctypes.c_int(-1)
        assert lib.gkr_selftest_hash8(_p(V), _p(L), _p(O), ctypes.byref(used)) == 0
        assert used.value in (0, 1)
        got = from_limbs(O)
        for k in range(8):
            assert got[k] == mimc7.multi_hash(vals[k][3 - lens[k]:]), (trial, k, lens[k])


_HASH8_CHILD = r"""
import ctypes, random, sys
import numpy as np
sys.path.insert(0, %r)
from gkr_amd import _native as N
from gkr_amd.field import MODULUS as P, to_limbs, from_limbs
lib = N.lib()
rng = random.Random(11)
for trial in range(6):
    lens = [rng.randint(0, 3) for _ in range(8)]
    vals = [[rng.randrange(P) for _ in range(3)] for _ in range(8)]
    if trial == 0:
        vals[0], vals[1], lens[0], lens[1] = [0, 0, 0], [P - 1, P - 1, P - 1], 3, 3
    V = to_limbs([x for row in vals for x in row])
    L = np.asarray(lens, dtype=np.uint32)
    O = np.zeros((8, 4), dtype=np.uint64)
    used = ctypes.c_int(-1)
    assert lib.gkr_selftest_hash8(V.ctypes.data_as(ctypes.c_void_p), L.ctypes.data_as(ctypes.c_void_p), O.ctypes.data_as(ctypes.c_void_p), ctypes.byref(used)) == 0
    assert used.value == 0
    print(lens, [row for row in vals], [int(x) for x in from_limbs(O)])
"""


@pytest.mark.parametrize("env", [{"GKR_NO_IFMA": "1"}, {"GKR_NO_IFMA": "1", "GKR_NO_ADX": "1"}], ids=["mulx_adx", "portable"])
def test_single_transcript_hash_matches_oracle(env):
    """one transcript's MiMC7 on the calling thread (batch 1 and chunk tails): the mulx / adcx / adox code of
    mimc_adx.cpp where the CPU has it, and the portable 4 x 64-bit code -- the knobs are read once per process"""
    import ast
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _HASH8_CHILD % root], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("[")]
    assert len(lines) == 6
    for line in lines:
        lens, vals, got = ast.literal_eval("(" + line.replace("] [", "], [") + ")")
        for k in range(8):
            assert got[k] == mimc7.multi_hash(vals[k][3 - lens[k]:]), (env, k)


def test_host_pass_matches_big_integers():
    """the host's share of a multi-round pass (round coefficients from the sub-block sums, vector lengths, MiMC7
    challenges, binding, the next fold's weights): scalar and IFMA-lane forms against Python integers and the
    oracle's hash, every lane count and pass depth, zero linear coefficients and edge values included"""
    lib = N.lib()
    rng = random.Random(21)
    for count in (1, 2, 3, 7, 8, 9, 16):
        for J in (1, 2, 3, 5):
            for final in (False, True):
                sums = [[rng.randrange(P) for _ in range(32)] for _ in range(count)]
                sums[0][:1 << J] = [rng.randrange(P)] * (1 << J)          # every c1 of lane 0 is zero
                if count > 1:
                    sums[1][:1 << J] = [P - 1, 0] * (1 << (J - 1))
                flen = [rng.choice((1, 2)) for _ in range(count)]
                S = to_limbs([x for row in sums for x in row])
                FL = np.asarray(flen, dtype=np.uint32)
                C0, C1, R = (np.zeros((J * count, 4), dtype=np.uint64) for _ in range(3))
                L = np.zeros(J * count, dtype=np.uint32)
                W = np.zeros((count * 32, 4), dtype=np.uint64)
                used = ctypes.c_int(-1)
                rc = lib.gkr_selftest_host_pass(_p(S), ctypes.c_int(count), ctypes.c_int(J), _p(FL) if final else None, _p(C0), _p(C1),
                                                _p(L), _p(R), _p(W), ctypes.byref(used))
                assert rc == 0, (count, J, final, rc)
                c0, c1, r, w_got = from_limbs(C0), from_limbs(C1), from_limbs(R), from_limbs(W)
                for k in range(count):
                    row = list(sums[k][:1 << J])
                    w = [1]
                    for t in range(J):
                        half = 1 << (J - t - 1)
                        lo, hi = sum(row[:half]) % P, sum(row[half:2 * half]) % P
                        d = (hi - lo) % P
                        ln = flen[k] if final and t == J - 1 else (1 if d == 0 else 2)
                        want_r = mimc7.multi_hash([d, lo][2 - ln:])
                        i = t * count + k
                        assert (c0[i], c1[i], int(L[i]), r[i]) == (lo, d, ln, want_r), (count, J, final, k, t)
                        row = [(row[b] + (row[half + b] - row[b]) * want_r) % P for b in range(half)]
                        w = [x for v in w for x in ((v * (1 - want_r)) % P, (v * want_r) % P)]
                    assert w_got[k * 32:k * 32 + (1 << J)] == w, (count, J, final, k)
    bad = to_limbs([0] * 32)
    bad[3] = 0xFFFFFFFFFFFFFFFF
    out = [np.zeros((1, 4), dtype=np.uint64) for _ in range(3)]
    assert lib.gkr_selftest_host_pass(_p(bad), ctypes.c_int(1), ctypes.c_int(1), None, _p(out[0]), _p(out[1]),
                                      _p(np.zeros(1, dtype=np.uint32)), _p(out[2]), _p(np.zeros((32, 4), dtype=np.uint64)), None) \
        == N.GKR_ERR_NON_CANONICAL


def test_host_tail_pass_forms_the_records_a_device_pass_would():
    """capi_layer.hip's host tail: the fold of the previous pass's variables and the cross sums of the next rounds, as sums of
    products with one reduction each (fr64.h wide_mac / wide_reduce) -- against plain integer arithmetic, with random tables,
    tables of r - 1 (the largest sums) and of zeros"""
    lib = N.lib()
    rng = random.Random(606)
    for m, jp, J, kind in ((6, 3, 3, "random"), (9, 3, 3, "random"), (5, 0, 3, "random"), (5, 3, 2, "random"), (3, 0, 3, "random"),
                           (12, 3, 3, "max"), (9, 3, 3, "max"), (4, 2, 2, "zero"), (6, 0, 1, "random"), (8, 3, 3, "random")):
        n = 1 << m
        draw = {"random": lambda: rng.randrange(P), "max": lambda: P - 1, "zero": lambda: 0}[kind]
        Wt, Xt, Yt = ([draw() for _ in range(n)] for _ in range(3))
        w = [draw() if kind != "zero" else rng.randrange(P) for _ in range(1 << jp)]
        mf = m - jp
        fold = lambda T: [sum(w[b] * T[(b << mf) + i] for b in range(1 << jp)) % P for i in range(1 << mf)] if jp else list(T)
        Wf, Xf, Yf = fold(Wt), fold(Xt), fold(Yt)
        S = (1 << mf) >> J
        want = [0] * 72
        for a in range(1 << J):
            for b in range(1 << J):
                want[a * 8 + b] = sum(Wf[a * S + i] * Xf[b * S + i] for i in range(S)) % P
            want[64 + a] = sum(Yf[a * S + i] for i in range(S)) % P
        rec = np.zeros((72, 4), dtype=np.uint64)
        T, Wl = to_limbs(Wt + Xt + Yt), to_limbs(w)          # (kept alive across the call)
        rc = lib.gkr_selftest_host_tail(_p(T), ctypes.c_int(m), ctypes.c_int(jp), _p(Wl) if jp else None, ctypes.c_int(J), _p(rec), None)
        assert rc == N.GKR_OK, (m, jp, J, kind)
        assert from_limbs(rec) == want, (m, jp, J, kind)
    bad, zeros, rec = to_limbs([0] * 24), to_limbs([0] * 24), np.zeros((72, 4), dtype=np.uint64)
    bad[5] = 0xFFFFFFFFFFFFFFFF
    assert lib.gkr_selftest_host_tail(_p(bad), ctypes.c_int(3), ctypes.c_int(0), None, ctypes.c_int(3), _p(rec), None) == N.GKR_ERR_NON_CANONICAL
    assert lib.gkr_selftest_host_tail(_p(zeros), ctypes.c_int(3), ctypes.c_int(0), None, ctypes.c_int(4), _p(rec), None) == N.GKR_ERR_INVALID


def test_host_prod_pass_matches_a_direct_product_sumcheck():
    """the host's share of a product pass of the layer sumcheck: the cross-sum matrix of random tables W, X, Y is built
    here, and the coefficients / challenges / weights the library derives from it alone (scalar and IFMA lanes) must
    equal a sumcheck of sum_t W(t) X(t) + Y(t) run directly on the tables, round by round"""
    lib = N.lib()
    rng = random.Random(33)
    for count in (1, 3, 8, 11, 16):
        for J in (1, 2, 3):
            m = J + rng.randint(0, 2)                       # tables of 2^m entries; the pass covers the first J variables
            S = 1 << (m - J)
            tabs = [[[rng.randrange(P) for _ in range(1 << m)] for _ in range(3)] for _ in range(count)]
            if count > 1:
                tabs[1][1] = [0] * (1 << m)                  # X = 0: every c2 is zero
            recs = []
            for Wt, Xt, Yt in tabs:
                rec = [0] * 72
                for a in range(1 << J):
                    for b in range(1 << J):
                        rec[a * 8 + b] = sum(Wt[a * S + i] * Xt[b * S + i] for i in range(S)) % P
                    rec[64 + a] = sum(Yt[a * S + i] for i in range(S)) % P
                recs += rec
            vl = [[rng.choice((2, 3)) for _ in range(count)] for _ in range(J)]
            R = to_limbs(recs)
            VL = np.asarray([x for row in vl for x in row], dtype=np.uint32)
            C2, LIN, C0, RR = (np.zeros((J * count, 4), dtype=np.uint64) for _ in range(4))
            Wout = np.zeros((count * 8, 4), dtype=np.uint64)
            used = ctypes.c_int(-1)
            rc = lib.gkr_selftest_host_prod_pass(_p(R), ctypes.c_int(count), ctypes.c_int(J), _p(VL), _p(C2), _p(LIN), _p(C0), _p(RR),
                                                 _p(Wout), ctypes.byref(used))
            assert rc == 0, (count, J, rc)
            c2, lin, c0, rr, w_got = from_limbs(C2), from_limbs(LIN), from_limbs(C0), from_limbs(RR), from_limbs(Wout)
            for k, (Wt, Xt, Yt) in enumerate(tabs):
                Wt, Xt, Yt = list(Wt), list(Xt), list(Yt)
                w = [1]
                for t in range(J):
                    h = len(Wt) // 2
                    want_c0 = sum(Wt[i] * Xt[i] + Yt[i] for i in range(h)) % P
                    g1 = sum(Wt[h + i] * Xt[h + i] + Yt[h + i] for i in range(h)) % P
                    want_c2 = sum((Wt[h + i] - Wt[i]) * (Xt[h + i] - Xt[i]) for i in range(h)) % P
                    want_lin = (g1 - want_c0 - want_c2) % P
                    want_r = mimc7.multi_hash([want_c2, want_lin, want_c0][3 - vl[t][k]:])
                    i = t * count + k
                    assert (c2[i], lin[i], c0[i], rr[i]) == (want_c2, want_lin, want_c0, want_r), (count, J, m, k, t)
                    Wt, Xt, Yt = ([(T[j] + (T[h + j] - T[j]) * want_r) % P for j in range(h)] for T in (Wt, Xt, Yt))
                    w = [x for v in w for x in ((v * (1 - want_r)) % P, (v * want_r) % P)]
                assert w_got[k * 8:k * 8 + (1 << J)] == w, (count, J, k)


def test_non_canonical_inputs_are_rejected():
    lib = N.lib()
    bad = np.full((1, 4), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
    ok = to_limbs([5])
    O = np.zeros((1, 4), dtype=np.uint64)
    assert lib.gkr_selftest_mul(_p(bad), _p(ok), _p(O)) == N.GKR_ERR_NON_CANONICAL
    exactly_r = to_limbs([0])
    for j in range(4):
        exactly_r[0, j] = (P >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    assert lib.gkr_mimc7_multi_hash(_p(exactly_r), ctypes.c_size_t(1), None, _p(O)) == N.GKR_ERR_NON_CANONICAL


def test_library_mimc7_matches_public_vectors_and_oracle():
    lib = N.lib()
    O = np.zeros((1, 4), dtype=np.uint64)
    for i in (0, 1, 2, 90):
        assert lib.gkr_mimc7_constant(ctypes.c_int(i), _p(O)) == 0
        assert from_limbs(O)[0] == mimc7.CTS[i]
    assert lib.gkr_mimc7_hash(_p(to_limbs([1])), _p(to_limbs([2])), _p(O)) == 0
    assert from_limbs(O)[0] == 0x176C6EEFC3FDF8D6136002D8E6F7A885BBD1C4E3957B93DDC1EC3AE7859F1A08
    assert gkr_amd.multi_hash([]) == 0
    assert gkr_amd.multi_hash([12, 45, 78, 41]) == 0x284BC1F34F335933A23A433B6FF3EE179D682CD5E5E2FCDD2D964AFA85104BEB
    rng = random.Random(4)
    for n in (1, 2, 3):
        xs = [rng.randrange(P) for _ in range(n)]
        assert gkr_amd.multi_hash(xs) == mimc7.multi_hash(xs)
    assert gkr_amd.multi_hash([7], key=9) == mimc7.multi_hash([7], 9)


def test_proof_sizes_host_only():
    lib = N.lib()
    k = np.asarray([1, 2, 2], dtype=np.uint32)
    dummy = (ctypes.c_void_p * 2)(0, 0)
    desc = N.CircuitDesc(2, k.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), dummy, dummy, dummy)
    s = N.ProofSizes()
    assert lib.gkr_proof_sizes(ctypes.byref(desc), ctypes.byref(s)) == 0
    assert (s.rounds, s.q_slots, s.z_values, s.d_coeffs, s.input_coeffs) == (8, 6, 5, 2, 4)
    k0 = np.asarray([1, 0], dtype=np.uint32)
    desc = N.CircuitDesc(1, k0.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), dummy, dummy, dummy)
    assert lib.gkr_proof_sizes(ctypes.byref(desc), ctypes.byref(s)) == N.GKR_ERR_DEGENERATE


def test_pass_schedule_of_the_plain_sumcheck():
    """Host logic of the multi-round passes: the rounds of the passes add up to n; a fold pass over more than 512
    outputs leaves at least 64 entries per sub-block of its sums (whole wave tiles) and never 1024 or 2048 entries;
    the first pass of a large table covers five rounds (three with the v_mad_u64_u32 fold)."""
    import ctypes
    import numpy as np
    from gkr_amd import _native as N
    L = N.lib()
    for mfma in (1, 0):
        for n in range(1, 33):
            rounds = np.zeros(64, dtype=np.uint32)
            count = ctypes.c_size_t()
            assert L.gkr_selftest_pass_schedule(ctypes.c_int(n), ctypes.c_int(mfma), rounds.ctypes.data_as(ctypes.c_void_p),
                                                ctypes.c_size_t(64), ctypes.byref(count)) == 0
            r = [int(x) for x in rounds[:count.value]]
            assert sum(r) == n and all(1 <= j <= (5 if mfma else 3) for j in r)
            m = n
            for i, j in enumerate(r):
                if i > 0 and (1 << m) > 512:          # this table came out of a fold pass whose sums have 2^j sub-blocks
                    assert m - j >= 6
                m -= j
                if mfma and m > 9:
                    assert m not in (10, 11)          # a fold leaves <= 512 entries (one-block kernel) or >= 4096
            if n >= 17:
                assert r[0] == (5 if mfma else 3)
    assert [int(x) for x in _schedule(L, 20, 1)] == [5, 3, 5, 5, 2]


def _schedule(L, n, mfma):
    import ctypes
    import numpy as np
    rounds = np.zeros(64, dtype=np.uint32)
    count = ctypes.c_size_t()
    assert L.gkr_selftest_pass_schedule(ctypes.c_int(n), ctypes.c_int(mfma), rounds.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(64),
                                        ctypes.byref(count)) == 0
    return rounds[:count.value]


def test_limb_widening_for_the_modular_all_reduce():
    """gkr_fr_widen / gkr_fr_narrow (host only): field elements as eight 32-bit limbs in int64, so that an ordinary
    integer SUM all-reduce followed by narrow is the modular sum over ranks (RCCL has no modular sum)."""
    import random
    from gkr_amd import parallel
    from gkr_amd.field import from_limbs, to_limbs
    rng = random.Random(31)
    vals = [P - 1, P - 2, 0, 1, (1 << 253) + 12345] + [rng.randrange(P) for _ in range(40)]
    w = parallel.widen(to_limbs(vals))
    assert w.shape == (len(vals), 8) and w.dtype == np.int64 and int(w.max()) < (1 << 32) and int(w.min()) >= 0
    assert from_limbs(parallel.narrow(w)) == vals
    for ranks in (2, 8, 1000, (1 << 31) - 1):
        assert from_limbs(parallel.narrow(w * ranks)) == [(v * ranks) % P for v in vals]
    other = parallel.widen(to_limbs(list(reversed(vals))))
    assert from_limbs(parallel.narrow(w + other)) == [(a + b) % P for a, b in zip(vals, reversed(vals))]
    bad = w.copy()
    bad[3, 2] = -1
    with pytest.raises(parallel.GkrError):
        parallel.narrow(bad)


def test_all_reduce_hook_round_trip_in_process():
    """The ctypes callback the library calls for a sum over ranks, driven here directly (no GPU): three logical
    ranks, in-process sum."""
    import ctypes
    import threading
    from gkr_amd import parallel
    from gkr_amd.field import from_limbs, to_limbs
    world = 3
    coll = parallel.ThreadedSum(world)
    parts = [[(P - 1 - 7 * r) % P, 1000 + r, 0] for r in range(world)]
    outs = [None] * world

    def run(r):
        hook, errors = parallel.make_allreduce_hook(coll.for_rank(r))
        buf = to_limbs(parts[r])
        assert hook(None, buf.ctypes.data_as(ctypes.c_void_p), len(parts[r])) == 0 and not errors
        outs[r] = from_limbs(buf)
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    want = [sum(col) % P for col in zip(*parts)]
    assert outs == [want] * world


def test_gate_ranges_partition_the_layer():
    from gkr_amd import parallel
    for k_i, world in ((0, 1), (0, 4), (5, 3), (10, 8), (3, 16)):
        spans = [parallel.gate_range(k_i, r, world) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == 1 << k_i
        assert all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(world - 1))


@pytest.mark.parametrize("k", [1, 2, 3, 5, 7])
def test_line_restriction_by_binding_equals_the_monomial_expansion(k):
    """q(t) = W(b + t (c - b)) (reduce_multiple_polynomial, poly.rs:469-500): the library binds the variables one by
    one on the evaluation table; the oracle expands every monomial as the reference does.  Same coefficients, same
    length (1 + largest degree of a stored monomial), also when W lacks variables or is constant."""
    from oracle import dense
    rng = random.Random(50 + k)
    n = 1 << k
    tables = [[rng.randrange(P) for _ in range(n)], [7] * n, [0] * n, [(i >> (k - 1)) + 3 for i in range(n)],
              [bin(i).count("1") % 2 for i in range(n)], [rng.randrange(3) for _ in range(n)]]
    for w in tables:
        b = [rng.randrange(P) for _ in range(k)]
        c = [rng.randrange(P) for _ in range(k)]
        out = np.zeros((k + 1, 4), dtype=np.uint64)
        ln = ctypes.c_uint32()
        rc = N.lib().gkr_selftest_line_restriction(ctypes.c_int(k), _p(to_limbs(w)), _p(to_limbs(b)), _p(to_limbs(c)), _p(out),
                                                  ctypes.byref(ln))
        assert rc == 0
        want = dense.line_restriction(b, c, w, k)
        got = from_limbs(out)
        assert got[k + 1 - ln.value:] == want and all(v == 0 for v in got[:k + 1 - ln.value])


# ---- the limits and the knobs: one table each, checked against the code -------------------------------------------------

def _header_limits():
    import re
    text = open(os.path.join(REPO, "include", "gkr_amd.h")).read()
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(GKR_MAX_[A-Z_]+)\s+(\d+)", text)}


def test_limits_table_matches_the_code():
    """include/gkr_amd.h's limits are the ones the entry points enforce (no device needed: gkr_proof_sizes validates the
    circuit description the way gkr_prove does)."""
    import ctypes
    import numpy as np
    lim = _header_limits()
    assert set(lim) == {"GKR_MAX_K_NEXT", "GKR_MAX_K_I", "GKR_MAX_K_NEXT_DEVICE_TRANSCRIPT", "GKR_MAX_MLE_N", "GKR_MAX_BATCH"}
    assert lim["GKR_MAX_K_NEXT"] >= 20, "the reference's compiler emits layers far beyond 2^14 values (convert.rs:10-11, 171-186)"
    lib = N.lib()

    def sizes_rc(ks):
        karr = np.asarray(ks, dtype=np.uint32)
        L = len(ks) - 1
        dummy = (ctypes.c_void_p * L)(*([1] * L))     # never dereferenced by gkr_proof_sizes
        desc = N.CircuitDesc(L, karr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), dummy, dummy, dummy)
        out = N.ProofSizes()
        return lib.gkr_proof_sizes(ctypes.byref(desc), ctypes.byref(out)), out
    rc, out = sizes_rc([3, lim["GKR_MAX_K_NEXT"], 5])
    assert rc == 0 and out.rounds == 2 * (lim["GKR_MAX_K_NEXT"] + 5) and out.input_coeffs == 32
    assert sizes_rc([3, lim["GKR_MAX_K_NEXT"] + 1, 5])[0] == N.GKR_ERR_INVALID
    assert sizes_rc([lim["GKR_MAX_K_I"], 4])[0] == 0
    assert sizes_rc([lim["GKR_MAX_K_I"] + 1, 4])[0] == N.GKR_ERR_INVALID
    assert sizes_rc([3, 0])[0] == N.GKR_ERR_DEGENERATE
    lib.gkr_exchange_limbs.restype = ctypes.c_size_t
    assert lib.gkr_exchange_limbs(ctypes.c_int(lim["GKR_MAX_K_NEXT"])) == ((2 << lim["GKR_MAX_K_NEXT"]) + 1) * 8
    assert lib.gkr_exchange_limbs(ctypes.c_int(lim["GKR_MAX_K_NEXT"] + 1)) == 0
    lib.gkr_exchange_limbs_mle.restype = ctypes.c_size_t
    assert lib.gkr_exchange_limbs_mle(ctypes.c_int(lim["GKR_MAX_MLE_N"] + 3), ctypes.c_int(3), ctypes.c_int(1)) > 0
    assert lib.gkr_exchange_limbs_mle(ctypes.c_int(lim["GKR_MAX_MLE_N"] + 4), ctypes.c_int(3), ctypes.c_int(1)) == 0
    # the limits are spelled in the sources by name, not as literals scattered over the file
    import glob
    capi = "".join(open(f).read() for f in glob.glob(os.path.join(REPO, "gkr_amd", "csrc", "*capi*")) if f.endswith((".hip", ".h")))
    for name in lim:
        assert name in capi or name == "GKR_MAX_BATCH", name


def test_every_switch_is_in_the_one_table():
    """csrc/options.h is the one place the library reads the environment: no other source calls getenv; INTEGRATION.md
    section 4 lists exactly the table's options (by name and variable) and the process switches options.h declares; and no
    switch that produces wrong results exists in the shipped library."""
    import glob
    import re
    from gkr_amd.prover import options
    csrc = os.path.join(REPO, "gkr_amd", "csrc")
    for path in glob.glob(os.path.join(csrc, "*")):
        if path.endswith((".hip", ".h", ".cpp")) and os.path.basename(path) != "options.h":
            assert "getenv" not in open(path).read(), "%s reads the environment itself" % path
    table = options()
    assert len(table) >= 20 and len({name for name, _, _ in table}) == len(table)
    opt_text = open(os.path.join(csrc, "options.h")).read()
    process = set(re.findall(r"^//   (GKR_[A-Z0-9_]+|LOCAL_WORLD_SIZE)\s", opt_text, flags=re.M))
    used = set()
    for path in glob.glob(os.path.join(csrc, "*")):
        if path.endswith((".hip", ".h", ".cpp")):
            used |= set(re.findall(r'process_(?:switch|int)\("([A-Z0-9_]+)"', open(path).read()))
    assert used == process, "process switches used %s, declared in options.h %s" % (sorted(used), sorted(process))
    doc_text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    section = doc_text[doc_text.index("## 4. Options and process switches"):]
    documented = set(re.findall(r"`(GKR_[A-Z0-9_]+)`", section)) - {"GKR_TRANSCRIPT_DEVICE"}
    live = {env for _, env, _ in table} | (process - {"LOCAL_WORLD_SIZE"})
    retired = {"GKR_GROUP_SPLIT", "GKR_STAGGER", "GKR_WAIT_MODE", "GKR_HELP_FLAT"}     # named in the "retired" paragraph
    assert live - documented == set(), "undocumented: %s" % sorted(live - documented)
    assert documented - live - retired == set(), "documented but gone: %s" % sorted(documented - live - retired)
    for name, _, _ in table:
        assert "`%s`" % name in section, name
    # the shipped library holds no experiment switch
    blob = open(N.LIB_PATH, "rb").read()
    assert b"GKR_DEBUG_SEG" not in blob and b"GKR_DEBUG_" not in blob.replace(b"GKR_DEBUG_TIMING", b"")


def test_context_options_without_a_device():
    """The option table is reachable without a GPU (names, variables, texts); set / get need a context."""
    from gkr_amd.prover import options
    names = [n for n, _, _ in options()]
    for must in ("rounds_per_pass", "no_mfma_fold", "gate_groups_min_k", "prove_many_lockstep", "no_circuit_cache"):
        assert must in names
    lib = N.lib()
    import ctypes
    v = ctypes.c_longlong()
    assert lib.gkr_ctx_get_option(None, b"rounds_per_pass", ctypes.byref(v)) == N.GKR_ERR_INVALID
    assert lib.gkr_ctx_set_option(None, b"rounds_per_pass", ctypes.c_longlong(3)) == N.GKR_ERR_INVALID


def test_rccl_exchange_without_a_device_is_an_error_not_a_crash():
    """gkr_exchange_rccl_* (the library-owned collective): librccl is loaded on first use; on a box without a GPU every entry
    point comes back with a status and a text, never a crash and never a fallback."""
    import ctypes
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = N.lib()
    lib.gkr_exchange_rccl_error.restype = ctypes.c_char_p
    h = ctypes.c_void_p()
    rc = lib.gkr_exchange_rccl_create(ctypes.c_int(0), ctypes.create_string_buffer(128), ctypes.c_int(0), ctypes.c_int(1), ctypes.c_size_t(64), ctypes.byref(h))
    assert rc == N.GKR_ERR_NO_DEVICE and not h.value and lib.gkr_exchange_rccl_error()
    assert lib.gkr_exchange_rccl_create(ctypes.c_int(0), None, ctypes.c_int(0), ctypes.c_int(1), ctypes.c_size_t(64), ctypes.byref(h)) == N.GKR_ERR_INVALID
    assert lib.gkr_exchange_rccl_create(ctypes.c_int(0), ctypes.create_string_buffer(128), ctypes.c_int(2), ctypes.c_int(2), ctypes.c_size_t(64), ctypes.byref(h)) == N.GKR_ERR_INVALID
    lib.gkr_exchange_rccl_dev.restype = ctypes.c_void_p
    assert lib.gkr_exchange_rccl_dev(None) is None
    lib.gkr_exchange_rccl_destroy.restype = None
    lib.gkr_exchange_rccl_destroy(None)
    ctx = ctypes.c_void_p()
    assert lib.gkr_ctx_create_multi((ctypes.c_int * 2)(0, 1), ctypes.c_int(2), ctypes.byref(ctx)) == N.GKR_ERR_NO_DEVICE
    assert lib.gkr_ctx_create_multi(None, ctypes.c_int(2), ctypes.byref(ctx)) == N.GKR_ERR_INVALID
