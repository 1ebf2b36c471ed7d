"""The reference's own argument types at `prove`, and the library's verifier (ctypes over csrc/dropin.cpp; host only).

prover::prove(&GKRCircuit, &Input) (rust/src/gkr/prover.rs:6-9) is handed each layer's wiring as 0/1 wire vectors
`gate || left || right` (Layer.wire, rust/src/gkr.rs:35-51; built at rust/src/convert.rs:715-767) and each layer's values
as a term list [coeff, e_1 .. e_k] (Input.w, gkr.rs:21-33; get_multi_ext, rust/src/gkr/poly.rs:502-536):

    layer_from_wires(k_i, k_next, add_wire, mult_wire) -> Layer          gkr_layer_from_wires
    values_from_terms(terms, k) -> [2^k ints]                            gkr_values_from_terms (inverse of get_multi_ext)
    terms_from_coeffs(coeff_limbs, k) -> term list                       gkr_terms_from_coeffs (Proof.d / Proof.input_func)
    prove_reference_types(ctx, layers_as_wires, input_w) -> Proof        prove() on exactly those types
    verify_native(circuit, arrays | Proof) -> (accept, layer, check)     gkr_verify (python/gkr.py:202-231 in C++)
"""

import ctypes
from typing import List, Sequence

import numpy as np

from . import _native as N
from .field import from_limbs, to_limbs
from .prover import GKRCircuit, GkrError, Layer, Proof, _proof_bufs, _ptr

VERIFY_CHECKS = {0: "ok", 1: "shape", 2: "non-canonical element", 3: "z[0] != 0", 4: "round sum", 5: "round challenge", 6: "final claim",
                 7: "r*", 8: "next z", 9: "input layer"}


def _rows(vectors: Sequence[Sequence[int]], width: int) -> np.ndarray:
    flat = [x for v in vectors for x in v]
    if len(flat) != len(vectors) * width:
        raise GkrError(N.GKR_ERR_INVALID, "every row needs %d elements" % width)
    return to_limbs(flat) if flat else np.zeros((0, 4), dtype=np.uint64)


def layer_from_wires(k_i: int, k_next: int, add_wire, mult_wire) -> Layer:
    """Layer.wire = (add_wire, mult_wire): lists of 0/1 vectors of length k_i + 2 k_next -> the gate arrays the ABI takes."""
    a, m = _rows(add_wire, k_i + 2 * k_next), _rows(mult_wire, k_i + 2 * k_next)
    g = 1 << k_i
    gt, l, r = np.zeros(g, dtype=np.uint8), np.zeros(g, dtype=np.uint32), np.zeros(g, dtype=np.uint32)
    rc = N.lib().gkr_layer_from_wires(ctypes.c_int(k_i), ctypes.c_int(k_next), _ptr(a), ctypes.c_size_t(len(add_wire)), _ptr(m),
                                      ctypes.c_size_t(len(mult_wire)), _ptr(gt), _ptr(l), _ptr(r))
    if rc:
        raise GkrError(rc, "gkr_layer_from_wires")
    return Layer(k_i, gt, l, r)


def values_from_terms(terms, k: int) -> List[int]:
    """A multilinear term list [coeff, e_1 .. e_k] (Input.w[i]) -> its 2^k evaluations, variable 1 = most significant bit."""
    t = _rows(terms, k + 1)
    out = np.zeros((1 << k, 4), dtype=np.uint64)
    rc = N.lib().gkr_values_from_terms(ctypes.c_int(k), _ptr(t), ctypes.c_size_t(len(terms)), _ptr(out))
    if rc:
        raise GkrError(rc, "gkr_values_from_terms")
    return from_limbs(out)


def terms_from_coeffs(coeff_limbs, k: int):
    """2^k monomial coefficients (gkr_proof_buf.d_coeffs / input_coeffs) -> the reference's term list (non-zero terms)."""
    c = np.ascontiguousarray(coeff_limbs, dtype=np.uint64).reshape(1 << k, 4)
    n = ctypes.c_size_t()
    lib = N.lib()
    rc = lib.gkr_terms_from_coeffs(ctypes.c_int(k), _ptr(c), None, ctypes.c_size_t(0), ctypes.byref(n))
    if rc:
        raise GkrError(rc, "gkr_terms_from_coeffs")
    out = np.zeros((max(1, n.value) * (k + 1), 4), dtype=np.uint64)
    rc = lib.gkr_terms_from_coeffs(ctypes.c_int(k), _ptr(c), _ptr(out), ctypes.c_size_t(n.value), ctypes.byref(n))
    if rc:
        raise GkrError(rc, "gkr_terms_from_coeffs")
    flat = from_limbs(out[:n.value * (k + 1)])
    return [flat[i * (k + 1):(i + 1) * (k + 1)] for i in range(n.value)]


def prove_reference_types(ctx, wires, input_w, k_list) -> Proof:
    """prover::prove on the types the reference hands it: wires[i] = (add_wire, mult_wire) of layer i, input_w = Input.w[depth]
    (the input layer's term list), k_list = GKRCircuit::get_k_list()."""
    layers = [layer_from_wires(k_list[i], k_list[i + 1], *wires[i]) for i in range(len(wires))]
    return ctx.prove(GKRCircuit(layers, k_list[-1]), values_from_terms(input_w, k_list[-1]))


def _arrays_of_proof(proof: Proof):
    """A decoded Proof -> the nine arrays of one gkr_proof_buf (batch axis of 1)."""
    ks = proof.k
    L = len(ks) - 1
    rounds = sum(2 * ks[i + 1] for i in range(L))
    sc = np.zeros((1, rounds, 3, 4), dtype=np.uint64)
    sl = np.zeros((1, rounds), dtype=np.uint32)
    sr = np.zeros((1, rounds, 4), dtype=np.uint64)
    q = np.zeros((1, sum(ks[i + 1] + 1 for i in range(L)), 4), dtype=np.uint64)
    ql = np.zeros((1, L), dtype=np.uint32)
    z = np.zeros((1, max(1, sum(ks)), 4), dtype=np.uint64)
    rr = np.zeros((1, L, 4), dtype=np.uint64)
    ro = qo = 0
    for i in range(L):
        k = ks[i + 1]
        for j in range(2 * k):
            g = proof.sumcheck_proofs[i][j]
            if not 1 <= len(g) <= 3:
                raise GkrError(N.GKR_ERR_INVALID, "a round vector of %d elements" % len(g))
            sl[0, ro + j] = len(g)
            sc[0, ro + j, 3 - len(g):] = to_limbs(g)
        sr[0, ro:ro + 2 * k] = to_limbs(proof.sumcheck_r[i])
        qi = proof.q[i]
        if not 1 <= len(qi) <= k + 1:
            raise GkrError(N.GKR_ERR_INVALID, "q of %d elements" % len(qi))
        ql[0, i] = len(qi)
        q[0, qo + k + 1 - len(qi):qo + k + 1] = to_limbs(qi)
        ro += 2 * k
        qo += k + 1
    zo = 0
    for i in range(L + 1):
        if ks[i]:
            z[0, zo:zo + ks[i]] = to_limbs(proof.z[i])
        zo += ks[i]
    rr[0] = to_limbs(proof.r)

    def coeffs(terms, k):
        out = [0] * (1 << k)
        for t in terms:
            m = 0
            for e in t[1:]:
                m = (m << 1) | int(e)
            out[m] = (out[m] + t[0])
        return to_limbs(out)[None]
    return [sc, sl, sr, q, ql, z, rr, coeffs(proof.d, ks[0]), coeffs(proof.input_func, ks[-1])]


def verify_native(circuit: GKRCircuit, proof, index: int = 0, threads: int = 0):
    """gkr_verify on one proof: `proof` is a Proof, or the nine raw output arrays of prove_batch_raw(all_arrays=True) /
    a prove_many item (then `index` picks the proof).  -> (accept, failed_layer, failed_check)."""
    from .prover import Context
    arrays = _arrays_of_proof(proof) if isinstance(proof, Proof) else [np.ascontiguousarray(a) for a in proof]
    if isinstance(proof, Proof):
        index = 0
    bufs = _proof_bufs(arrays, arrays[0].shape[0])
    desc, alive = Context._circuit_desc(None, circuit)
    accept, layer, check = ctypes.c_int(0), ctypes.c_uint32(0), ctypes.c_uint32(0)
    rc = N.lib().gkr_verify(ctypes.byref(desc), ctypes.byref(bufs[index]), ctypes.c_int(threads), ctypes.byref(accept), ctypes.byref(layer),
                            ctypes.byref(check))
    if rc:
        raise GkrError(rc, "gkr_verify")
    return bool(accept.value), int(layer.value), int(check.value)
