"""ctypes front-end of the plain-C dense oracle (oracle/c/libogkr.so).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Field elements travel as
numpy uint64 arrays of shape (..., 4): little-endian limbs of the canonical
value.
"""

import ctypes
import os
import subprocess

import numpy as np

from .field import P

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "c", "libogkr.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "c")], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.ogkr_max_threads.restype = ctypes.c_int
        # OpenMP's default is one thread per visible CPU; a container whose cgroup quota is smaller (the GPU box
        # shows 256 CPUs under a 16-CPU quota) then spends its time throttled
        if hasattr(_lib, "ogkr_set_threads"):
            _lib.ogkr_set_threads(ctypes.c_int(_quota_cpus()))
    return _lib


def _quota_cpus():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, n)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def to_limbs(values):
    """list of Python ints -> uint64 array (n, 4)."""
    out = np.empty((len(values), 4), dtype=np.uint64)
    for i, v in enumerate(values):
        v %= P
        for j in range(4):
            out[i, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


def from_limbs(arr):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [sum(int(arr[i, j]) << (64 * j) for j in range(4)) for i in range(arr.shape[0])]


def max_threads():
    return lib().ogkr_max_threads()


def usable_threads():
    """Threads the process can really run (affinity and cgroup quota), what the library's loops default to."""
    return max(1, min(_quota_cpus(), max_threads()))


def fr_mul(a, b):
    A, B = to_limbs([a]), to_limbs([b])
    O = np.zeros((1, 4), dtype=np.uint64)
    lib().ogkr_fr_mul(_p(A), _p(B), _p(O))
    return from_limbs(O)[0]


def multi_hash(values, key=0):
    A = to_limbs(list(values)) if len(values) else np.zeros((0, 4), dtype=np.uint64)
    K = to_limbs([key])
    O = np.zeros((1, 4), dtype=np.uint64)
    lib().ogkr_multi_hash(_p(A), ctypes.c_size_t(len(values)), _p(K), _p(O))
    return from_limbs(O)[0]


def mimc7_constant(i):
    O = np.zeros((1, 4), dtype=np.uint64)
    lib().ogkr_mimc7_constant(ctypes.c_int(i), _p(O))
    return from_limbs(O)[0]


def fill_table(count, seed):
    T = np.empty((count, 4), dtype=np.uint64)
    lib().ogkr_fill_table(_p(T), ctypes.c_size_t(count), ctypes.c_uint64(seed))
    return T


def sumcheck_mle_raw(table_limbs, n, threads=0):
    """-> (coeffs (n,2,4) uint64 right-aligned, lens (n,) uint32, r (n,4) uint64)."""
    table_limbs = np.ascontiguousarray(table_limbs, dtype=np.uint64)
    assert table_limbs.shape == (1 << n, 4)
    C = np.zeros((n, 2, 4), dtype=np.uint64)
    L = np.zeros(n, dtype=np.uint32)
    R = np.zeros((n, 4), dtype=np.uint64)
    rc = lib().ogkr_sumcheck_mle(_p(table_limbs), ctypes.c_int(n), _p(C), _p(L), _p(R), ctypes.c_int(threads if threads > 0 else usable_threads()))
    if rc:
        raise ValueError("ogkr_sumcheck_mle rc=%d" % rc)
    return C, L, R


def sumcheck_mle_inplace_raw(table_limbs, n, threads=0):
    """sumcheck_mle_raw on the caller's array, which is overwritten (no copy: 2^30 entries are 32 GiB)."""
    assert table_limbs.flags["C_CONTIGUOUS"] and table_limbs.dtype == np.uint64 and table_limbs.shape == (1 << n, 4)
    C = np.zeros((n, 2, 4), dtype=np.uint64)
    L = np.zeros(n, dtype=np.uint32)
    R = np.zeros((n, 4), dtype=np.uint64)
    rc = lib().ogkr_sumcheck_mle_inplace(_p(table_limbs), ctypes.c_int(n), _p(C), _p(L), _p(R), ctypes.c_int(threads if threads > 0 else usable_threads()))
    if rc:
        raise ValueError("ogkr_sumcheck_mle_inplace rc=%d" % rc)
    return C, L, R


def sumcheck_mle(table, n, threads=0):
    C, L, R = sumcheck_mle_raw(to_limbs(table), n, threads)
    proof = [from_limbs(C[j])[2 - int(L[j]):] for j in range(n)]
    return proof, from_limbs(R)


def _gates(gate_type, left, right):
    return (np.ascontiguousarray(gate_type, dtype=np.uint8), np.ascontiguousarray(left, dtype=np.uint32),
            np.ascontiguousarray(right, dtype=np.uint32))


def sumcheck_layer_raw(k_i, k_next, gate_type, left, right, z_limbs, w_limbs, threads=0):
    gt, l, r = _gates(gate_type, left, right)
    v = 2 * k_next
    C = np.zeros((v, 3, 4), dtype=np.uint64)
    L = np.zeros(v, dtype=np.uint32)
    R = np.zeros((v, 4), dtype=np.uint64)
    z_limbs = np.ascontiguousarray(z_limbs, dtype=np.uint64).reshape(-1, 4)
    w_limbs = np.ascontiguousarray(w_limbs, dtype=np.uint64)
    rc = lib().ogkr_sumcheck_layer(ctypes.c_int(k_i), ctypes.c_int(k_next), _p(gt), _p(l), _p(r), _p(z_limbs),
                                   _p(w_limbs), _p(C), _p(L), _p(R), ctypes.c_int(threads if threads > 0 else usable_threads()))
    if rc:
        raise ValueError("ogkr_sumcheck_layer rc=%d" % rc)
    return C, L, R


def sumcheck_layer_lin_raw(k_i, k_next, gate_type, left, right, z_limbs, w_limbs, threads=0):
    """The linear-time twin (ogkr_sumcheck_layer_lin): same outputs as sumcheck_layer_raw, any k_next <= 28."""
    gt, l, r = _gates(gate_type, left, right)
    v = 2 * k_next
    C = np.zeros((v, 3, 4), dtype=np.uint64)
    L = np.zeros(v, dtype=np.uint32)
    R = np.zeros((v, 4), dtype=np.uint64)
    z_limbs = np.ascontiguousarray(z_limbs, dtype=np.uint64).reshape(-1, 4)
    w_limbs = np.ascontiguousarray(w_limbs, dtype=np.uint64)
    assert w_limbs.shape == (1 << k_next, 4) and len(gt) == 1 << k_i
    rc = lib().ogkr_sumcheck_layer_lin(ctypes.c_int(k_i), ctypes.c_int(k_next), _p(gt), _p(l), _p(r), _p(z_limbs),
                                       _p(w_limbs), _p(C), _p(L), _p(R), ctypes.c_int(threads if threads > 0 else usable_threads()))
    if rc:
        raise ValueError("ogkr_sumcheck_layer_lin rc=%d" % rc)
    return C, L, R


def sumcheck_layer_lin(k_i, k_next, gate_type, left, right, z, w, threads=0):
    zl = to_limbs(z) if len(z) else np.zeros((0, 4), dtype=np.uint64)
    C, L, R = sumcheck_layer_lin_raw(k_i, k_next, gate_type, left, right, zl, to_limbs(w), threads)
    proof = [from_limbs(C[j])[3 - int(L[j]):] for j in range(2 * k_next)]
    return proof, from_limbs(R)


def sumcheck_layer(k_i, k_next, gate_type, left, right, z, w, threads=0):
    zl = to_limbs(z) if len(z) else np.zeros((0, 4), dtype=np.uint64)
    C, L, R = sumcheck_layer_raw(k_i, k_next, gate_type, left, right, zl, to_limbs(w), threads)
    proof = [from_limbs(C[j])[3 - int(L[j]):] for j in range(2 * k_next)]
    return proof, from_limbs(R)


def predicate_tables(k_i, k_next, gate_type, left, right, z):
    gt, l, r = _gates(gate_type, left, right)
    n = 1 << (2 * k_next)
    A = np.zeros((n, 4), dtype=np.uint64)
    M = np.zeros((n, 4), dtype=np.uint64)
    zl = to_limbs(z) if len(z) else np.zeros((0, 4), dtype=np.uint64)
    rc = lib().ogkr_predicate_tables(ctypes.c_int(k_i), ctypes.c_int(k_next), _p(gt), _p(l), _p(r), _p(zl), _p(A), _p(M))
    if rc:
        raise ValueError("ogkr_predicate_tables rc=%d" % rc)
    return A, M


def layer_eval_raw(gate_type, left, right, prev_limbs):
    gt, l, r = _gates(gate_type, left, right)
    prev_limbs = np.ascontiguousarray(prev_limbs, dtype=np.uint64)
    out = np.zeros((len(gt), 4), dtype=np.uint64)
    lib().ogkr_layer_eval(ctypes.c_size_t(len(gt)), _p(gt), _p(l), _p(r), _p(prev_limbs), _p(out))
    return out


def line_restriction_raw(b_limbs, c_limbs, w_limbs, k):
    """-> (q (k+1,4) right-aligned highest first, length)."""
    O = np.zeros((k + 1, 4), dtype=np.uint64)
    ln = ctypes.c_uint32(0)
    B = np.ascontiguousarray(b_limbs, dtype=np.uint64).reshape(-1, 4)
    Cc = np.ascontiguousarray(c_limbs, dtype=np.uint64).reshape(-1, 4)
    Wl = np.ascontiguousarray(w_limbs, dtype=np.uint64)
    rc = lib().ogkr_line_restriction(ctypes.c_int(k), _p(B), _p(Cc), _p(Wl), _p(O), ctypes.byref(ln))
    if rc:
        raise ValueError("ogkr_line_restriction rc=%d" % rc)
    return O, int(ln.value)


def mobius_raw(w_limbs, k):
    """evaluation table -> monomial coefficients (get_multi_ext, poly.rs:502-536), limbs in, limbs out."""
    out = np.array(w_limbs, dtype=np.uint64, copy=True).reshape(1 << k, 4)
    lib().ogkr_mobius(_p(out), ctypes.c_int(k))
    return out


def prove_raw(layers, input_limbs, threads=0):
    """prover.rs:6-96 on limb arrays only (no Python integers per table entry): for circuits with wide layers, where
    the dense layer form cannot follow.  Uses the linear-time layer prover.  z[0] = 0 (prover.rs:16-21).
    -> dict of numpy arrays: per layer C (2k,3,4), L (2k,), R (2k,4), q (k+1,4), q_len; z list of (k_i,4); r (depth,4);
    values list of (2^k_i,4)."""
    vals = [np.ascontiguousarray(input_limbs, dtype=np.uint64).reshape(-1, 4)]
    for gt, l, r in reversed(layers):
        vals.append(layer_eval_raw(gt, l, r, vals[-1]))
    vals.reverse()
    ks = [max(0, (v.shape[0] - 1).bit_length()) for v in vals]
    z = [np.zeros((ks[0], 4), dtype=np.uint64)]
    Cs, Ls, Rs, qs, qlens, rstars = [], [], [], [], [], []
    for i, (gt, l, r) in enumerate(layers):
        kn = ks[i + 1]
        C, L, R = sumcheck_layer_lin_raw(ks[i], kn, gt, l, r, z[i], vals[i + 1], threads)
        Cs.append(C)
        Ls.append(L)
        Rs.append(R)
        q, qlen = line_restriction_raw(R[:kn], R[kn:], vals[i + 1], kn)
        qs.append(q)
        qlens.append(qlen)
        r_star = from_limbs(R[2 * kn - 1:2 * kn])[0]   # multi_hash(last round vector) = the last challenge (prover.rs:74-78)
        bs, cs = from_limbs(R[:kn]), from_limbs(R[kn:])
        z.append(to_limbs([(bi + (ci - bi) * r_star) % P for bi, ci in zip(bs, cs)]))
        rstars.append(r_star)
    return dict(C=Cs, L=Ls, R=Rs, q=qs, q_len=qlens, z=z, r=to_limbs(rstars), k=ks, values=vals)


def line_restriction(b, c, w, k):
    O = np.zeros((k + 1, 4), dtype=np.uint64)
    ln = ctypes.c_uint32(0)
    B = to_limbs(b) if k else np.zeros((0, 4), dtype=np.uint64)
    Cc = to_limbs(c) if k else np.zeros((0, 4), dtype=np.uint64)
    rc = lib().ogkr_line_restriction(ctypes.c_int(k), _p(B), _p(Cc), _p(to_limbs(w)), _p(O), ctypes.byref(ln))
    if rc:
        raise ValueError("ogkr_line_restriction rc=%d" % rc)
    return from_limbs(O)[k + 1 - ln.value:]


def prove(layers, input_values, z0=None, threads=0):
    """prover.rs:6-96 composed from the C pieces (same dict as dense.prove)."""
    from .mimc7 import multi_hash as py_hash  # tiny; the C hash is checked against it separately
    vals = [to_limbs(list(input_values))]
    for gt, l, r in reversed(layers):
        vals.append(layer_eval_raw(gt, l, r, vals[-1]))
    vals.reverse()
    ks = [max(0, (v.shape[0] - 1).bit_length()) for v in vals]
    z = [[0] * ks[0]] if z0 is None else [[x % P for x in z0]]
    sps, srs, qs, rstars = [], [], [], []
    for i, (gt, l, r) in enumerate(layers):
        kn = ks[i + 1]
        zl = to_limbs(z[i]) if ks[i] else np.zeros((0, 4), dtype=np.uint64)
        C, L, R = sumcheck_layer_raw(ks[i], kn, gt, l, r, zl, vals[i + 1], threads)
        sp = [from_limbs(C[j])[3 - int(L[j]):] for j in range(2 * kn)]
        sr = from_limbs(R)
        sps.append(sp)
        srs.append(sr)
        b_star, c_star = sr[:kn], sr[kn:]
        qs.append(line_restriction(b_star, c_star, from_limbs(vals[i + 1]), kn))
        r_star = multi_hash(sp[-1], 0)
        assert r_star == py_hash(sp[-1], 0)
        z.append([(bi + (ci - bi) * r_star) % P for bi, ci in zip(b_star, c_star)])
        rstars.append(r_star)
    return dict(sumcheck_proofs=sps, sumcheck_r=srs, q=qs, z=z, r=rstars, depth=len(layers) + 1, k=ks,
                values=[from_limbs(v) for v in vals])
