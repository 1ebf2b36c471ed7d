"""The proof as input signals of verifier.circom -- the artefact right after the hot path.

Mirror of the reference's `get_meta` (aggregator.rs:92-146), `modify_proof_for_circom` (:148-213),
`CircomInputProof::new_from_proof` (:20-82) and `write_aggregated_input` (file_utils.rs:49-67).  The
formatting itself is done by the library (`gkr_circom_meta`, `gkr_circom_input_json`, host-only C++ in
csrc/circom_input.cpp); this module re-encodes a `Proof` into the ABI's flat buffers and parses the result.
"""

import ctypes
import json
from typing import Dict, List, Sequence

import numpy as np

from . import _native as N
from .field import MODULUS, from_limbs, to_limbs
from .prover import GkrError, Proof


def _limbs(values):
    return to_limbs([v % MODULUS for v in values]) if len(values) else np.zeros((0, 4), dtype=np.uint64)


class _Encoded:
    """A Proof in the C ABI's layout (gkr_proof_buf, include/gkr_amd.h); keeps the arrays alive."""

    def __init__(self, proof: Proof):
        ks = list(proof.k)
        L = proof.depth - 1
        if len(ks) != L + 1 or len(proof.sumcheck_proofs) != L:
            raise GkrError(N.GKR_ERR_INVALID, "proof shape does not match its k list")
        rounds = sum(2 * ks[i + 1] for i in range(L))
        self.coeffs = np.zeros((max(rounds, 1), 3, 4), dtype=np.uint64)
        self.lens = np.zeros(max(rounds, 1), dtype=np.uint32)
        self.rs = np.zeros((max(rounds, 1), 4), dtype=np.uint64)
        self.q = np.zeros((max(sum(ks[i + 1] + 1 for i in range(L)), 1), 4), dtype=np.uint64)
        self.q_len = np.zeros(max(L, 1), dtype=np.uint32)
        self.z = np.zeros((max(sum(ks), 1), 4), dtype=np.uint64)
        self.r = _limbs(proof.r) if L else np.zeros((1, 4), dtype=np.uint64)
        row = qo = 0
        for i in range(L):
            k = ks[i + 1]
            for j, vec in enumerate(proof.sumcheck_proofs[i]):
                self.coeffs[row + j, 3 - len(vec):] = _limbs(vec)
                self.lens[row + j] = len(vec)
            self.rs[row:row + 2 * k] = _limbs(proof.sumcheck_r[i])
            qv = proof.q[i]
            self.q[qo + k + 1 - len(qv):qo + k + 1] = _limbs(qv)
            self.q_len[i] = len(qv)
            row += 2 * k
            qo += k + 1
        zo = 0
        for i in range(L + 1):
            if ks[i]:
                self.z[zo:zo + ks[i]] = _limbs(proof.z[i])
            zo += ks[i]
        self.d = self._coeff_table(proof.d, ks[0])
        self.inp = self._coeff_table(proof.input_func, ks[L])
        self.karr = np.asarray(ks, dtype=np.uint32)
        self.desc = N.CircuitDesc(L, self.karr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), None, None, None)
        self.buf = N.ProofBuf(*[a.ctypes.data for a in (self.coeffs, self.lens, self.rs, self.q, self.q_len, self.z, self.r,
                                                        self.d, self.inp)])

    @staticmethod
    def _coeff_table(terms, k):
        """term list [coeff, e_1..e_k] -> 2^k monomial coefficients (variable 1 = most significant index bit)."""
        vals = [0] * (1 << k)
        for t in terms:
            m = 0
            for e in t[1:]:
                m = (m << 1) | (int(e) & 1)
            vals[m] = (vals[m] + t[0]) % MODULUS
        return _limbs(vals)


def circom_meta(proof: Proof) -> List[int]:
    """get_meta (aggregator.rs:92-146): the VerifyGKR template arguments of one proof."""
    e = _Encoded(proof)
    out = np.zeros(8 + len(proof.k), dtype=np.uint32)
    count = ctypes.c_size_t()
    rc = N.lib().gkr_circom_meta(ctypes.byref(e.desc), ctypes.byref(e.buf), out.ctypes.data_as(ctypes.c_void_p),
                                 ctypes.c_size_t(len(out)), ctypes.byref(count))
    if rc:
        raise GkrError(rc, "gkr_circom_meta")
    return [int(x) for x in out[:count.value]]


def circom_input(proof: Proof, index: int) -> Dict[str, list]:
    """The seven padded arrays of decimal strings, keys suffixed with the proof's index."""
    e = _Encoded(proof)
    need = ctypes.c_size_t()
    rc = N.lib().gkr_circom_input_json(ctypes.byref(e.desc), ctypes.byref(e.buf), ctypes.c_int(index), None, ctypes.c_size_t(0),
                                       ctypes.byref(need))
    if rc:
        raise GkrError(rc, "gkr_circom_input_json")
    text = ctypes.create_string_buffer(need.value)
    rc = N.lib().gkr_circom_input_json(ctypes.byref(e.desc), ctypes.byref(e.buf), ctypes.c_int(index), text,
                                       ctypes.c_size_t(need.value), ctypes.byref(need))
    if rc:
        raise GkrError(rc, "gkr_circom_input_json")
    return json.loads(text.value.decode())


def _text_call(fn, *args):
    need = ctypes.c_size_t()
    rc = fn(*args, None, ctypes.c_size_t(0), ctypes.byref(need))
    if rc:
        raise GkrError(rc, fn.__name__)
    buf = ctypes.create_string_buffer(need.value)
    rc = fn(*args, buf, ctypes.c_size_t(need.value), ctypes.byref(need))
    if rc:
        raise GkrError(rc, fn.__name__)
    return buf.value.decode()


def verifier_source(metas: Sequence[Sequence[int]]) -> str:
    """The circom text modify_circom_file renders for a list of proofs (aggregator.rs:216-290)."""
    flat = np.asarray([x for m in metas for x in m], dtype=np.uint32)
    lens = (ctypes.c_size_t * max(len(metas), 1))(*[len(m) for m in metas])
    return _text_call(N.lib().gkr_circom_verifier_source, flat.ctypes.data_as(ctypes.c_void_p), lens, ctypes.c_size_t(len(metas)))


def modify_circom_file(circuit_text: str, metas: Sequence[Sequence[int]]) -> str:
    """modify_circom_file (aggregator.rs:215-314) as a text function: the circuit with the verifier components of the
    previous round's proofs injected.  (The reference writes it to aggregated.circom and runs circom on it.)"""
    return _text_call(N.lib().gkr_circom_inject, circuit_text.encode(), verifier_source(metas).encode())


def prove_step(ctx, r1cs, witnesses, require_zero_output=True):
    """The proving step of one aggregation round (aggregator.rs:341-355, 399-416): compile the R1CS into its
    <= 20 layered circuits (convert_r1cs_wtns_gkr) and prove every (circuit, input) pair.  The reference maps
    prover::prove over the pairs of ONE witness with a rayon par_iter; with several witnesses of the same R1CS
    (BASELINE configs[3]: 64 inputs of one circom circuit) the sub-circuits coincide, so each sub-circuit's proofs
    for all witnesses advance together (one gkr_prove_batch each), and the sub-circuits are proven side by side
    (gkr_prove_many -- the par_iter).  -> proofs[w][j] for witness w and sub-circuit j."""
    from .field import as_limbs
    layered = r1cs.compile()
    try:
        wl = [as_limbs(w) for w in witnesses]
        work = [(layered.circuit(j), np.stack([layered.input_values_raw(j, w) for w in wl])) for j in range(len(layered))]
        per_circuit = ctx.prove_many(work, require_zero_output)     # the sub-circuits side by side (gkr_prove_many)
        return [[per_circuit[j][w] for j in range(len(layered))] for w in range(len(witnesses))]
    finally:
        layered.close()


class ProvingStep:
    """prove_step with the compile done once and no decoding of the proofs into Python integers: what bench.py
    times.  inputs_for gathers the witnesses into every sub-circuit's input layer (calculate_input, convert.rs:796-
    810); prove_raw runs gkr_prove_batch per sub-circuit and returns the challenge arrays."""

    def __init__(self, r1cs):
        self._layered = r1cs.compile()
        self.circuits = [self._layered.circuit(j) for j in range(len(self._layered))]

    def inputs_for(self, witness_limbs):
        """witness_limbs: (W, n_wires, 4) uint64 -> one (W, 2^input_k, 4) array per sub-circuit."""
        return [np.stack([self._layered.input_values_raw(j, w) for w in witness_limbs]) for j in range(len(self.circuits))]

    def prove_raw(self, ctx, inputs):
        return [ctx.prove_batch_raw(c, x) for c, x in zip(self.circuits, inputs)]

    def prove_raw_concurrent(self, ctxs, inputs):
        """The sub-circuits proven from len(ctxs) threads, one context each (the reference's par_iter over the
        (circuit, input) pairs, aggregator.rs:350-355): a layer round is latency-bound (launch, tiny kernel, hand-off,
        hash), so independent sub-circuits in flight together fill each other's gaps.  Sub-circuits are dealt out
        largest first."""
        import threading
        order = sorted(range(len(self.circuits)), key=lambda j: -sum(self.circuits[j].get_k_list()))
        out, errs, lock = [None] * len(self.circuits), [], threading.Lock()
        busy = ctypes.c_int32(len(ctxs))   # threads still proving; the others lend themselves to them

        def work(ctx):
            try:
                while True:
                    with lock:
                        if not order or errs:
                            return
                        j = order.pop(0)
                    try:
                        out[j] = ctx.prove_batch_raw(self.circuits[j], inputs[j])
                    except Exception as e:
                        errs.append(e)
                        return
            finally:
                with lock:
                    busy.value -= 1
                # the sub-circuits differ in depth: a thread that is out of work takes pieces of the host work (hash
                # calls, line restrictions) of the contexts still proving (gkr_host_help_while), until all are done
                N.lib().gkr_host_help_while(ctypes.byref(busy))
        threads = [threading.Thread(target=work, args=(c,)) for c in ctxs[1:]]
        for t in threads:
            t.start()
        work(ctxs[0])
        for t in threads:
            t.join()
        if errs:
            raise errs[0]
        return out

    def prove_raw_many(self, ctx, inputs, max_concurrent=0):
        """The same step through gkr_prove_many: one call, the library's own threads and child contexts instead of
        interpreter threads (no GIL hand-offs, no thread start per call).  -> the challenge arrays, as prove_raw.

        `inputs` must be C-contiguous uint64 arrays (what inputs_for returns): the prepared item list points INTO them,
        so that a caller proving step after step updates them in place; the list is kept for as long as the same array
        objects and the same (open) context come back.  The returned arrays are the prepared list's output buffers:
        the next call overwrites them -- copy what must outlive it."""
        for x in inputs:
            if not (isinstance(x, np.ndarray) and x.dtype == np.uint64 and x.flags["C_CONTIGUOUS"]):
                raise ValueError("prove_raw_many needs C-contiguous uint64 input arrays (a converted copy would not see later updates)")
        handle = getattr(ctx, "_h", None)
        key = (id(ctx), handle.value if handle else None) + tuple(id(x) for x in inputs) + tuple(x.shape for x in inputs)
        if getattr(self, "_prepared_key", None) != key or handle is None or not handle.value:
            self._prepared = ctx.prepare_many(list(zip(self.circuits, inputs)))
            self._prepared_inputs = list(inputs)   # the key holds ids: keep the objects alive so that no id is reused
            self._prepared_key = key
        return [arrs[2] for arrs in ctx.prove_many_raw(self._prepared, max_concurrent)]

    @staticmethod
    def contexts_for(device, cpus, limit=12):
        """One context per concurrently proven sub-circuit, sized to the CPUs this process may use: every context's
        calling thread spins on its rounds' hand-off records, so contexts + workers must not exceed the cores
        (two are left to the runtime's own threads).  -> list of Context."""
        from .prover import Context
        n = max(1, min(limit, cpus - 2))
        each = max(1, (cpus - 2) // n)
        ctxs = [Context(device) for _ in range(n)]
        for c in ctxs:
            c.set_host_threads(each)
        return ctxs

    def close(self):
        self._layered.close()


def aggregated_input(circuit_input: Dict[str, object], proofs: Sequence[Proof]) -> Dict[str, object]:
    """write_aggregated_input (file_utils.rs:49-67): the circuit's own inputs plus every proof's signals."""
    out = dict(circuit_input)
    for i, p in enumerate(proofs):
        out.update(circom_input(p, i))
    return out
