"""Host-side mirror of the reference's prover interface over the C ABI.

Names follow the reference (rust/src/gkr.rs, rust/src/gkr/prover.rs,
rust/src/gkr/sumcheck.rs) so parity tests read like tests of the reference:

    reference (Rust)                                   here
    ------------------------------------------------   ---------------------------------
    GKRCircuit { layer: Vec<Layer>, input_k }          GKRCircuit(layers, input_k)
    Layer { k, add, mult, wire }                       Layer(k, gate_type, left, right)
    prover::prove(&circuit, &input) -> Proof           prove(circuit, input_values) -> Proof
    sumcheck::prove_sumcheck_opt(..., v)               prove_sumcheck_opt(layer, k_next, z, W)
    sumcheck::prove_sumcheck(g, v)                     prove_sumcheck(table, v)
    Mimc7::multi_hash(v, &Fr::from(0))                 multi_hash(values, key=0)

The reference stores a layer's wiring as term-list polynomials add_i / mult_i
plus 0/1 wire strings (convert.rs:715-774); both are functions of the gate
list (type, left operand, right operand), which is what this interface takes.
Everything numeric happens in libgkr_amd.so on the GPU; this module only
marshals.  Errors: the reference panics, here a GkrError carries the status.
"""

import ctypes
from dataclasses import dataclass, field as dc_field
from typing import List, Sequence

import numpy as np

from . import _native as N
from .field import MODULUS, as_limbs, from_limbs, to_limbs


class GkrError(RuntimeError):
    def __init__(self, status, message=""):
        self.status = status
        text = N.lib().gkr_strerror(status).decode()
        super().__init__("%s (status %d)%s" % (text, status, ": " + message if message else ""))


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@dataclass
class Layer:
    """One gate layer: gate g is add (0) or mult (1) of entries left[g], right[g]
    of the next layer (gkr.rs:35-51; wiring convert.rs:715-767)."""
    k: int
    gate_type: Sequence[int]
    left: Sequence[int]
    right: Sequence[int]

    def arrays(self):
        gt = np.ascontiguousarray(self.gate_type, dtype=np.uint8)
        l = np.ascontiguousarray(self.left, dtype=np.uint32)
        r = np.ascontiguousarray(self.right, dtype=np.uint32)
        if not (len(gt) == len(l) == len(r) == 1 << self.k):
            raise GkrError(N.GKR_ERR_INVALID, "layer with k=%d needs %d gates" % (self.k, 1 << self.k))
        return gt, l, r


@dataclass
class GKRCircuit:
    """gkr.rs:53-114."""
    layer: List[Layer]
    input_k: int

    def depth(self):
        return len(self.layer)

    def k(self, i):
        return self.input_k if i == len(self.layer) else self.layer[i].k

    def get_k_list(self):
        return [self.k(i) for i in range(self.depth() + 1)]


@dataclass
class Proof:
    """gkr.rs:7-19.  d / input_func are term lists [coeff, e_1..e_k] (order not
    significant: the reference emits them in HashMap order)."""
    sumcheck_proofs: List[List[List[int]]]
    sumcheck_r: List[List[int]]
    d: List[List[int]]
    q: List[List[int]]
    z: List[List[int]]
    r: List[int]
    depth: int
    input_func: List[List[int]]
    k: List[int]
    values: List[List[int]] = dc_field(default_factory=list, repr=False)


def multi_hash(values, key=0):
    """MiMC7-91 multi_hash on the host side of the library (no GPU needed)."""
    arr = to_limbs(values) if len(values) else np.zeros((0, 4), dtype=np.uint64)
    k = to_limbs([key])
    out = np.zeros((1, 4), dtype=np.uint64)
    rc = N.lib().gkr_mimc7_multi_hash(_ptr(arr), ctypes.c_size_t(len(values)), _ptr(k), _ptr(out))
    if rc:
        raise GkrError(rc)
    return from_limbs(out)[0]


def options():
    """The library's option table: [(name, environment variable, what it does)] (no GPU needed)."""
    lib = N.lib()
    for f in (lib.gkr_option_name, lib.gkr_option_doc, lib.gkr_option_env):
        f.restype = ctypes.c_char_p
        f.argtypes = [ctypes.c_int]
    lib.gkr_option_count.restype = ctypes.c_int
    return [(lib.gkr_option_name(i).decode(), lib.gkr_option_env(i).decode(), lib.gkr_option_doc(i).decode())
            for i in range(lib.gkr_option_count())]


def host_hash_us(length):
    """(us per hash in sixteen IFMA lanes, us per scalar hash) of a `length`-element round vector on one host thread."""
    a, b = ctypes.c_double(), ctypes.c_double()
    rc = N.lib().gkr_ubench_host_hash(ctypes.c_int(length), ctypes.byref(a), ctypes.byref(b))
    if rc:
        raise GkrError(rc)
    return a.value, b.value


def _proof_bufs(arrays, B):
    """The B gkr_proof_buf structures (nine pointers each) over arrays whose first axis is the proof: one address
    matrix, base + b * stride per array, built in numpy -- not B x 9 ctypes objects (0.5 - 1 ms of interpreter time per
    call, which the proving threads of ProvingStep.prove_raw_concurrent pay one after the other under the GIL)."""
    addr = np.empty((B, len(arrays)), dtype=np.uint64)
    for j, a in enumerate(arrays):
        addr[:, j] = a.ctypes.data + np.arange(B, dtype=np.uint64) * np.uint64(a.strides[0])
    return (N.ProofBuf * B).from_buffer(addr)


def _decode_proofs(arrays, ks):
    """The nine output arrays of gkr_prove_batch / one gkr_prove_many item (first axis = proof) -> Proof objects."""
    sc, sl, sr, q, ql, z, rr, dco, ico = arrays
    L = len(ks) - 1
    out = []
    for b in range(sc.shape[0]):
        proofs, rs, qs, zs = [], [], [], []
        ro = qo = 0
        for i in range(L):
            k = ks[i + 1]
            proofs.append([from_limbs(sc[b, ro + j])[3 - int(sl[b, ro + j]):] for j in range(2 * k)])
            rs.append(from_limbs(sr[b, ro:ro + 2 * k]))
            qs.append(from_limbs(q[b, qo:qo + k + 1])[k + 1 - int(ql[b, i]):])
            ro += 2 * k
            qo += k + 1
        zo = 0
        for i in range(L + 1):
            zs.append(from_limbs(z[b, zo:zo + ks[i]]) if ks[i] else [])
            zo += ks[i]
        out.append(Proof(sumcheck_proofs=proofs, sumcheck_r=rs, d=_terms_from_coeffs(dco[b], ks[0]), q=qs, z=zs,
                         r=from_limbs(rr[b]), depth=L + 1, input_func=_terms_from_coeffs(ico[b], ks[-1]), k=ks))
    return out


def _terms_from_coeffs(coeffs, k):
    """monomial-coefficient table -> reference term list (non-zero terms only)."""
    vals = from_limbs(coeffs)
    return [[c] + [(m >> (k - 1 - j)) & 1 for j in range(k)] for m, c in enumerate(vals) if c]


class Context:
    """One GPU context (one HIP stream).  Not thread-safe; use one per thread."""

    def __init__(self, device=0, devices=None):
        """device: the GPU of this context.  devices=[ids]: gkr_ctx_create_multi -- the context lives on ids[0] and
        prove_many deals its items over child contexts on all of them (one host process, several GPUs)."""
        self._h = ctypes.c_void_p()
        if devices is not None:
            ids = (ctypes.c_int * len(devices))(*devices)
            rc = N.lib().gkr_ctx_create_multi(ids, ctypes.c_int(len(devices)), ctypes.byref(self._h))
        else:
            rc = N.lib().gkr_ctx_create(ctypes.c_int(device), ctypes.byref(self._h))
        if rc:
            self._h = None
            raise GkrError(rc, "gkr_ctx_create(device=%d)" % device)

    def close(self):
        if self._h:
            N.lib().gkr_ctx_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise GkrError(rc, (N.lib().gkr_last_error(self._h) or b"").decode())

    # -- context knobs
    def device_name(self):
        buf = ctypes.create_string_buffer(256)
        self._check(N.lib().gkr_ctx_device_name(self._h, buf, ctypes.c_size_t(256)))
        return buf.value.decode()

    def device_count(self):
        """Devices prove_many deals over (1 unless created with devices=[...])."""
        N.lib().gkr_ctx_device_count.restype = ctypes.c_int
        return int(N.lib().gkr_ctx_device_count(self._h))

    def set_transcript(self, mode):
        self._check(N.lib().gkr_ctx_set_transcript(self._h, ctypes.c_int(mode)))

    def set_option(self, name, value):
        """One of the library's switches (csrc/options.h; gkr_amd.options() lists them) for THIS context only."""
        N.lib().gkr_ctx_set_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_longlong]
        self._check(N.lib().gkr_ctx_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name):
        N.lib().gkr_ctx_get_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_longlong)]
        v = ctypes.c_longlong()
        self._check(N.lib().gkr_ctx_get_option(self._h, name.encode(), ctypes.byref(v)))
        return int(v.value)

    def set_host_threads(self, threads):
        """Host threads the context may use for the transcript, the caller included (0 = default)."""
        self._check(N.lib().gkr_ctx_set_host_threads(self._h, ctypes.c_int(threads)))

    def profile(self, enable=True):
        """0/False off, 1/True every kernel, 2 the bandwidth-bound kernels only (see gkr_amd.h)."""
        self._check(N.lib().gkr_ctx_profile(self._h, ctypes.c_int(int(enable))))

    def profile_reset(self):
        self._check(N.lib().gkr_ctx_profile_reset(self._h))

    def profile_get(self, kernel):
        n = ctypes.c_uint64()
        ms = ctypes.c_double()
        b = ctypes.c_double()
        self._check(N.lib().gkr_ctx_profile_get(self._h, kernel.encode(), ctypes.byref(n), ctypes.byref(ms), ctypes.byref(b)))
        return dict(launches=n.value, total_ms=ms.value, bytes=b.value)

    def profile_samples(self, kernel, capacity=4096):
        """[(ms, algorithmic bytes)] of the single launches of `kernel` since the last reset."""
        ms = np.zeros(capacity, dtype=np.float64)
        by = np.zeros(capacity, dtype=np.float64)
        n = ctypes.c_size_t()
        self._check(N.lib().gkr_ctx_profile_samples(self._h, kernel.encode(), _ptr(ms), _ptr(by), ctypes.c_size_t(capacity),
                                                    ctypes.byref(n)))
        k = min(n.value, capacity)
        return list(zip(ms[:k].tolist(), by[:k].tolist()))

    # -- device memory
    def alloc(self, nbytes):
        p = ctypes.c_void_p()
        self._check(N.lib().gkr_device_alloc(self._h, ctypes.c_size_t(nbytes), ctypes.byref(p)))
        return p

    def free(self, dptr):
        self._check(N.lib().gkr_device_free(self._h, dptr))

    def upload(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self._check(N.lib().gkr_device_upload(self._h, dptr, _ptr(arr), ctypes.c_size_t(arr.nbytes)))

    def download(self, dptr, shape, dtype=np.uint64):
        out = np.empty(shape, dtype=dtype)
        self._check(N.lib().gkr_device_download(self._h, _ptr(out), dptr, ctypes.c_size_t(out.nbytes)))
        return out

    def fill_table(self, dptr, count, seed):
        self._check(N.lib().gkr_device_fill_table(self._h, dptr, ctypes.c_size_t(count), ctypes.c_uint64(seed)))

    def fill_shard(self, dptr, n, log2_shards, shard, seed):
        """Rank `shard`'s entries (gkr_sumcheck_mle_sharded_dev's layout) of the table fill_table(dptr, 2^n, seed) writes."""
        self._check(N.lib().gkr_device_fill_shard(self._h, dptr, ctypes.c_int(n), ctypes.c_int(log2_shards), ctypes.c_int(shard),
                                                  ctypes.c_uint64(seed)))

    def synchronize(self):
        self._check(N.lib().gkr_device_synchronize(self._h))

    def ceilings(self, nbytes=2 << 30):
        """The box's own ceilings, measured now: {"copy_GBps", "read_GBps", "modmul_per_sec"} (gkr_ubench_ceilings)."""
        c, r, m = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        self._check(N.lib().gkr_ubench_ceilings(self._h, ctypes.c_size_t(nbytes), ctypes.byref(c), ctypes.byref(r), ctypes.byref(m)))
        return {"copy_GBps": c.value, "read_GBps": r.value, "modmul_per_sec": m.value}

    # -- plain multilinear sumcheck (prove_sumcheck, sumcheck.rs:158-214)
    def sumcheck_mle_raw(self, table_limbs, n):
        table_limbs = np.ascontiguousarray(table_limbs, dtype=np.uint64)
        if table_limbs.shape != (1 << n, 4):
            raise GkrError(N.GKR_ERR_INVALID, "table must have 2^n rows")
        C = np.zeros((n, 2, 4), dtype=np.uint64)
        L = np.zeros(n, dtype=np.uint32)
        R = np.zeros((n, 4), dtype=np.uint64)
        self._check(N.lib().gkr_sumcheck_mle(self._h, _ptr(table_limbs), ctypes.c_int(n), _ptr(C), _ptr(L), _ptr(R)))
        return C, L, R

    def sumcheck_mle_batch_device(self, d_tables, n, batch, out=None):
        """out: (C, L, R) arrays of an earlier call to write into (a caller that proves batch after batch saves the
        page faults of fresh output arrays, which otherwise land inside the hash workers)."""
        if out is not None:
            C, L, R = out
            if C.shape != (batch, n, 2, 4) or L.shape != (batch, n) or R.shape != (batch, n, 4):
                raise GkrError(N.GKR_ERR_INVALID, "output arrays do not match (batch, n)")
        else:
            C = np.zeros((batch, n, 2, 4), dtype=np.uint64)
            L = np.zeros((batch, n), dtype=np.uint32)
            R = np.zeros((batch, n, 4), dtype=np.uint64)
        self._check(N.lib().gkr_sumcheck_mle_batch_device(self._h, d_tables, ctypes.c_int(n), ctypes.c_int(batch),
                                                          _ptr(C), _ptr(L), _ptr(R)))
        return C, L, R

    def prove_sumcheck(self, table, v):
        """(proof, r) like the reference: proof[j] is the round vector, highest degree first."""
        C, L, R = self.sumcheck_mle_raw(as_limbs(table), v)
        return [from_limbs(C[j])[2 - int(L[j]):] for j in range(v)], from_limbs(R)

    # -- layer sumcheck (prove_sumcheck_opt, sumcheck.rs:36-156)
    def sumcheck_layer_raw(self, layer: Layer, k_next, z, W):
        gt, l, r = layer.arrays()
        zl = as_limbs(z) if layer.k else np.zeros((0, 4), dtype=np.uint64)
        wl = as_limbs(W)
        if zl.shape[0] != layer.k or wl.shape[0] != (1 << max(k_next, 0)):
            raise GkrError(N.GKR_ERR_INVALID, "z needs k_i entries and W 2^k_next entries")
        v = 2 * max(k_next, 0)
        C = np.zeros((max(v, 1), 3, 4), dtype=np.uint64)
        L = np.zeros(max(v, 1), dtype=np.uint32)
        R = np.zeros((max(v, 1), 4), dtype=np.uint64)
        self._check(N.lib().gkr_sumcheck_layer(self._h, ctypes.c_int(layer.k), ctypes.c_int(k_next), _ptr(gt), _ptr(l),
                                               _ptr(r), _ptr(zl), _ptr(wl), _ptr(C), _ptr(L), _ptr(R)))
        return C[:v], L[:v], R[:v]

    def prove_sumcheck_opt(self, layer: Layer, k_next, z, W):
        C, L, R = self.sumcheck_layer_raw(layer, k_next, z, W)
        return [from_limbs(C[j])[3 - int(L[j]):] for j in range(2 * k_next)], from_limbs(R)

    def predicate_tables(self, layer: Layer, k_next, z):
        gt, l, r = layer.arrays()
        zl = as_limbs(z) if layer.k else np.zeros((0, 4), dtype=np.uint64)
        n = 1 << (2 * max(k_next, 0))
        A = np.zeros((n, 4), dtype=np.uint64)
        M = np.zeros((n, 4), dtype=np.uint64)
        self._check(N.lib().gkr_predicate_tables(self._h, ctypes.c_int(layer.k), ctypes.c_int(k_next), _ptr(gt), _ptr(l),
                                                 _ptr(r), _ptr(zl), _ptr(A), _ptr(M)))
        return A, M

    def layer_eval(self, layer: Layer, prev):
        gt, l, r = layer.arrays()
        pl = as_limbs(prev)
        out = np.zeros((len(gt), 4), dtype=np.uint64)
        self._check(N.lib().gkr_layer_eval(self._h, ctypes.c_size_t(len(gt)), _ptr(gt), _ptr(l), _ptr(r), _ptr(pl),
                                           ctypes.c_size_t(pl.shape[0]), _ptr(out)))
        return out

    # -- full proof (prover::prove, prover.rs:6-96)
    def prove(self, circuit: GKRCircuit, input_values, require_zero_output=False) -> Proof:
        return self.prove_batch(circuit, [input_values], require_zero_output)[0]

    def prove_batch(self, circuit: GKRCircuit, inputs, require_zero_output=False) -> List[Proof]:
        """Proofs of one circuit for several witnesses, advanced together on the GPU (gkr_prove_batch)."""
        L = circuit.depth()
        B = len(inputs)
        ks = circuit.get_k_list()
        karr = np.asarray(ks, dtype=np.uint32)
        gates = [lay.arrays() for lay in circuit.layer]
        gt_p = (ctypes.c_void_p * L)(*[g[0].ctypes.data for g in gates])
        l_p = (ctypes.c_void_p * L)(*[g[1].ctypes.data for g in gates])
        r_p = (ctypes.c_void_p * L)(*[g[2].ctypes.data for g in gates])
        desc = N.CircuitDesc(L, karr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), gt_p, l_p, r_p)
        sizes = N.ProofSizes()
        rc = N.lib().gkr_proof_sizes(ctypes.byref(desc), ctypes.byref(sizes))
        if rc:
            raise GkrError(rc, "gkr_proof_sizes")
        n_in = 1 << ks[-1]
        inp = np.empty((B, n_in, 4), dtype=np.uint64)
        for b, vals in enumerate(inputs):
            a = as_limbs(vals)
            if a.shape[0] != n_in:
                raise GkrError(N.GKR_ERR_INVALID, "input layer needs 2^input_k values")
            inp[b] = a
        sc = np.zeros((B, sizes.rounds, 3, 4), dtype=np.uint64)
        sl = np.zeros((B, sizes.rounds), dtype=np.uint32)
        sr = np.zeros((B, sizes.rounds, 4), dtype=np.uint64)
        q = np.zeros((B, sizes.q_slots, 4), dtype=np.uint64)
        ql = np.zeros((B, L), dtype=np.uint32)
        z = np.zeros((B, max(sizes.z_values, 1), 4), dtype=np.uint64)
        rr = np.zeros((B, L, 4), dtype=np.uint64)
        dco = np.zeros((B, sizes.d_coeffs, 4), dtype=np.uint64)
        ico = np.zeros((B, sizes.input_coeffs, 4), dtype=np.uint64)
        bufs = _proof_bufs((sc, sl, sr, q, ql, z, rr, dco, ico), B)
        self._check(N.lib().gkr_prove_batch(self._h, ctypes.byref(desc), _ptr(inp), ctypes.c_int(B),
                                            ctypes.c_int(1 if require_zero_output else 0), bufs))
        out = _decode_proofs((sc, sl, sr, q, ql, z, rr, dco, ico), ks)
        return out

    def _circuit_desc(self, circuit: GKRCircuit):
        """-> (gkr_circuit_desc, objects that must outlive its use)"""
        L = circuit.depth()
        karr = np.asarray(circuit.get_k_list(), dtype=np.uint32)
        gates = [lay.arrays() for lay in circuit.layer]
        gt_p = (ctypes.c_void_p * L)(*[g[0].ctypes.data for g in gates])
        l_p = (ctypes.c_void_p * L)(*[g[1].ctypes.data for g in gates])
        r_p = (ctypes.c_void_p * L)(*[g[2].ctypes.data for g in gates])
        desc = N.CircuitDesc(L, karr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), gt_p, l_p, r_p)
        return desc, (karr, gates, gt_p, l_p, r_p)

    def prepare_many(self, work, require_zero_output=False):
        """work: [(GKRCircuit, inputs_limbs (B, 2^input_k, 4) uint64)] -> a prepared item list for prove_many_raw:
        circuit descriptions, output arrays and the gkr_prove_item array built once (an aggregation step proves the same
        circuits for every batch of inputs; only the input values change)."""
        keep, outs = [], []
        items = (N.ProveItem * len(work))()
        for i, (circuit, inputs_limbs) in enumerate(work):
            desc, alive = self._circuit_desc(circuit)
            sizes = N.ProofSizes()
            self._check(N.lib().gkr_proof_sizes(ctypes.byref(desc), ctypes.byref(sizes)))
            inp = np.ascontiguousarray(inputs_limbs, dtype=np.uint64)
            B, L = inp.shape[0], circuit.depth()
            arrs = [np.zeros((B, sizes.rounds, 3, 4), dtype=np.uint64), np.zeros((B, sizes.rounds), dtype=np.uint32),
                    np.zeros((B, sizes.rounds, 4), dtype=np.uint64), np.zeros((B, sizes.q_slots, 4), dtype=np.uint64),
                    np.zeros((B, L), dtype=np.uint32), np.zeros((B, max(sizes.z_values, 1), 4), dtype=np.uint64),
                    np.zeros((B, L, 4), dtype=np.uint64), np.zeros((B, sizes.d_coeffs, 4), dtype=np.uint64),
                    np.zeros((B, sizes.input_coeffs, 4), dtype=np.uint64)]
            bufs = _proof_bufs(arrs, B)
            items[i] = N.ProveItem(ctypes.cast(ctypes.pointer(desc), ctypes.c_void_p), inp.ctypes.data, B,
                                   1 if require_zero_output else 0, ctypes.cast(bufs, ctypes.c_void_p), 0)
            keep.append((desc, alive, inp, arrs, bufs))
            outs.append(arrs)
        return {"items": items, "keep": keep, "outs": outs}

    def prove_many_raw(self, prepared, max_concurrent=0):
        """gkr_prove_many on a prepare_many() list: every item proven by its own thread / child context of this context
        (the reference's par_iter over the (circuit, input) pairs).  -> per item the list of output arrays
        [coeffs, lens, challenges, q, q_len, z, r, d, input] (first axis = proof)."""
        items = prepared["items"]
        rc = N.lib().gkr_prove_many(self._h, items, ctypes.c_size_t(len(items)), ctypes.c_int(max_concurrent))
        self._check(rc)
        return prepared["outs"]

    def prove_many(self, work, require_zero_output=False, max_concurrent=0) -> List[List[Proof]]:
        """gkr_prove_many, decoded: work = [(GKRCircuit, inputs_limbs)] -> per item its proofs (one per witness)."""
        prepared = self.prepare_many(work, require_zero_output)
        outs = self.prove_many_raw(prepared, max_concurrent)
        return [_decode_proofs(arrs, circuit.get_k_list()) for arrs, (circuit, _) in zip(outs, work)]

    def prove_batch_raw(self, circuit: GKRCircuit, inputs_limbs, all_arrays=False, out=None):
        """gkr_prove_batch without decoding the outputs into Python ints (bench.py's proofs/sec leg; circuits with wide
        layers, whose d / input_func tables have 2^18 and more entries).
        inputs_limbs: (B, 2^input_k, 4) uint64.  Returns the challenge arrays (B, rounds, 4), or with all_arrays the nine
        output arrays [coeffs, lens, challenges, q, q_len, z, r, d, input] (first axis = proof)."""
        L = circuit.depth()
        B = inputs_limbs.shape[0]
        ks = circuit.get_k_list()
        karr = np.asarray(ks, dtype=np.uint32)
        gates = [lay.arrays() for lay in circuit.layer]
        gt_p = (ctypes.c_void_p * L)(*[g[0].ctypes.data for g in gates])
        l_p = (ctypes.c_void_p * L)(*[g[1].ctypes.data for g in gates])
        r_p = (ctypes.c_void_p * L)(*[g[2].ctypes.data for g in gates])
        desc = N.CircuitDesc(L, karr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), gt_p, l_p, r_p)
        sizes = N.ProofSizes()
        self._check(N.lib().gkr_proof_sizes(ctypes.byref(desc), ctypes.byref(sizes)))
        inp = np.ascontiguousarray(inputs_limbs, dtype=np.uint64)
        # (out: the nine arrays of an earlier call with the same shapes, reused -- a proof with a 2^20-value input layer
        # carries 40 MiB of coefficients, and fresh pages for them cost more than the proof)
        arrs = out if out is not None else [
            np.zeros((B, sizes.rounds, 3, 4), dtype=np.uint64), np.zeros((B, sizes.rounds), dtype=np.uint32),
            np.zeros((B, sizes.rounds, 4), dtype=np.uint64), np.zeros((B, sizes.q_slots, 4), dtype=np.uint64),
            np.zeros((B, L), dtype=np.uint32), np.zeros((B, max(sizes.z_values, 1), 4), dtype=np.uint64),
            np.zeros((B, L, 4), dtype=np.uint64), np.zeros((B, sizes.d_coeffs, 4), dtype=np.uint64),
            np.zeros((B, sizes.input_coeffs, 4), dtype=np.uint64)]
        bufs = _proof_bufs(arrs, B)
        self._check(N.lib().gkr_prove_batch(self._h, ctypes.byref(desc), _ptr(inp), ctypes.c_int(B), ctypes.c_int(0), bufs))
        return arrs if (all_arrays or out is not None) else arrs[2]


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def prove(circuit: GKRCircuit, input_values, require_zero_output=False) -> Proof:
    """prover::prove (prover.rs:6-9) on cuda:0 / HIP device 0."""
    return default_context().prove(circuit, input_values, require_zero_output)


def prove_sumcheck_opt(layer: Layer, k_next, z, W):
    return default_context().prove_sumcheck_opt(layer, k_next, z, W)


def prove_sumcheck(table, v):
    return default_context().prove_sumcheck(table, v)
