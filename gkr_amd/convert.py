"""Host-side mirror of the step in front of the prover: R1CS + witness -> layered GKR circuits
(rust/src/convert.rs) over the C ABI (gkr_r1cs_* / gkr_wtns_* / gkr_layered_* in include/gkr_amd.h; the work is
done by gkr_amd/csrc/r1cs.cpp, host-only).

    reference (Rust)                                            here
    ---------------------------------------------------------   -----------------------------------------
    R1csFile::<32>::read(File::open(path))                      R1cs.parse(bytes) / R1cs.read(path)
    WtnsFile::<32>::read(File::open(path))                      read_wtns(bytes)
    convert_r1cs_wtns_gkr(r1cs, wtns, sym)                      convert_r1cs_wtns_gkr(r1cs, witness)
        -> (Vec<GKRCircuit>, Vec<Input>, Output)                    -> (circuits, input value vectors)
    compile(convert_constraints_to_nodes(&r1cs))                R1cs.compile() -> Layered

The reference's `Input` carries every layer's values as term-list polynomials (calculate_input,
convert.rs:787-849); here the input LAYER's values are what crosses the boundary (gkr_prove evaluates the layers
on the GPU and checks that output 0 is zero).  There is no circom in this image, so R1cs.build / serialize and
write_wtns exist to make fixtures.
"""

import ctypes
from typing import List, Sequence, Tuple

import numpy as np

from . import _native as N
from .field import as_limbs, from_limbs, to_limbs
from .prover import GKRCircuit, GkrError, Layer


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _check(rc, what):
    if rc:
        raise GkrError(rc, what)


class R1cs:
    """A parsed / built rank-1 constraint system over BN254 Fr.  A constraint is (A, B, C), each a list of
    (coefficient, wire) -- the tuple order of the reference's r1cs-file `Constraint` (convert.rs:368)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def parse(cls, data: bytes):
        h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(data, len(data))
        _check(N.lib().gkr_r1cs_parse(buf, ctypes.c_size_t(len(data)), ctypes.byref(h)), "gkr_r1cs_parse")
        return cls(h)

    @classmethod
    def read(cls, path):
        with open(path, "rb") as f:
            return cls.parse(f.read())

    @classmethod
    def build(cls, n_wires, n_pub_out, n_pub_in, n_prv_in, constraints: Sequence[Tuple[list, list, list]]):
        counts, wires, coeffs = [], [], []
        for con in constraints:
            for lc in con:
                counts.append(len(lc))
                for coeff, wire in lc:
                    wires.append(wire)
                    coeffs.append(coeff)
        counts = np.asarray(counts, dtype=np.uint32).reshape(-1)
        wires = np.asarray(wires, dtype=np.uint32).reshape(-1)
        cl = to_limbs(coeffs) if coeffs else np.zeros((0, 4), dtype=np.uint64)
        h = ctypes.c_void_p()
        _check(N.lib().gkr_r1cs_build(ctypes.c_uint32(n_wires), ctypes.c_uint32(n_pub_out), ctypes.c_uint32(n_pub_in),
                                      ctypes.c_uint32(n_prv_in), ctypes.c_size_t(len(constraints)), _ptr(counts), _ptr(wires),
                                      _ptr(cl), ctypes.byref(h)), "gkr_r1cs_build")
        return cls(h)

    def info(self):
        out = N.R1csInfo()
        _check(N.lib().gkr_r1cs_info(self._h, ctypes.byref(out)), "gkr_r1cs_info")
        return {k: int(getattr(out, k)) for k, _ in N.R1csInfo._fields_}

    def constraints(self):
        inf = self.info()
        counts = np.zeros(3 * inf["n_constraints"], dtype=np.uint32)
        wires = np.zeros(max(inf["n_terms"], 1), dtype=np.uint32)
        coeffs = np.zeros((max(inf["n_terms"], 1), 4), dtype=np.uint64)
        _check(N.lib().gkr_r1cs_export(self._h, _ptr(counts), _ptr(wires), _ptr(coeffs)), "gkr_r1cs_export")
        vals, out, pos = from_limbs(coeffs), [], 0
        for i in range(inf["n_constraints"]):
            con = []
            for j in range(3):
                n = int(counts[3 * i + j])
                con.append([(vals[pos + t], int(wires[pos + t])) for t in range(n)])
                pos += n
            out.append(tuple(con))
        return out

    def serialize(self) -> bytes:
        need = ctypes.c_size_t()
        _check(N.lib().gkr_r1cs_serialize(self._h, None, ctypes.c_size_t(0), ctypes.byref(need)), "gkr_r1cs_serialize")
        buf = ctypes.create_string_buffer(need.value)
        _check(N.lib().gkr_r1cs_serialize(self._h, buf, ctypes.c_size_t(need.value), ctypes.byref(need)), "gkr_r1cs_serialize")
        return buf.raw

    def compile(self):
        h = ctypes.c_void_p()
        bad = ctypes.c_size_t(0)
        rc = N.lib().gkr_r1cs_compile(self._h, ctypes.byref(h), ctypes.byref(bad))
        if rc == N.GKR_ERR_UNSUPPORTED:
            raise GkrError(rc, "constraint %d has an empty A, B or C (the reference recurses without end, convert.rs:619-622)"
                           % bad.value)
        _check(rc, "gkr_r1cs_compile")
        return Layered(h)

    def close(self):
        if self._h:
            N.lib().gkr_r1cs_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Layered:
    """The <= 20 layered circuits of one R1CS (compile, convert.rs:154-358)."""

    def __init__(self, handle):
        self._h = handle

    def __len__(self):
        n = ctypes.c_uint32()
        _check(N.lib().gkr_layered_count(self._h, ctypes.byref(n)), "gkr_layered_count")
        return n.value

    def circuit(self, index) -> GKRCircuit:
        desc = N.CircuitDesc()
        _check(N.lib().gkr_layered_circuit(self._h, ctypes.c_uint32(index), ctypes.byref(desc)), "gkr_layered_circuit")
        L = desc.depth
        ks = [int(desc.k[i]) for i in range(L + 1)]
        layers = []
        for i in range(L):
            g = 1 << ks[i]
            gt = np.ctypeslib.as_array(ctypes.cast(desc.gate_type[i], ctypes.POINTER(ctypes.c_uint8)), shape=(g,)).copy()
            l = np.ctypeslib.as_array(ctypes.cast(desc.left[i], ctypes.POINTER(ctypes.c_uint32)), shape=(g,)).copy()
            r = np.ctypeslib.as_array(ctypes.cast(desc.right[i], ctypes.POINTER(ctypes.c_uint32)), shape=(g,)).copy()
            layers.append(Layer(ks[i], gt, l, r))
        return GKRCircuit(layers, ks[-1])

    def input_layer(self, index):
        """[("var", wire) | ("val", constant)] per slot of the input layer."""
        wire = ctypes.POINTER(ctypes.c_uint32)()
        const = ctypes.c_void_p()
        slots = ctypes.c_size_t()
        _check(N.lib().gkr_layered_input_layer(self._h, ctypes.c_uint32(index), ctypes.byref(wire), ctypes.byref(const),
                                               ctypes.byref(slots)), "gkr_layered_input_layer")
        n = slots.value
        w = np.ctypeslib.as_array(wire, shape=(n,))
        c = from_limbs(np.ctypeslib.as_array(ctypes.cast(const, ctypes.POINTER(ctypes.c_uint64)), shape=(n, 4)))
        return [("val", c[s]) if int(w[s]) == 0xFFFFFFFF else ("var", int(w[s])) for s in range(n)]

    def input_values_raw(self, index, witness_limbs):
        wl = np.ascontiguousarray(witness_limbs, dtype=np.uint64).reshape(-1, 4)
        slots = ctypes.c_size_t()
        _check(N.lib().gkr_layered_input_layer(self._h, ctypes.c_uint32(index), None, None, ctypes.byref(slots)),
               "gkr_layered_input_layer")
        out = np.zeros((slots.value, 4), dtype=np.uint64)
        _check(N.lib().gkr_layered_input_values(self._h, ctypes.c_uint32(index), _ptr(wl), ctypes.c_size_t(wl.shape[0]), _ptr(out)),
               "gkr_layered_input_values")
        return out

    def input_values(self, index, witness: Sequence[int]) -> List[int]:
        return from_limbs(self.input_values_raw(index, as_limbs(witness)))

    def close(self):
        if self._h:
            N.lib().gkr_layered_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def read_wtns(data: bytes) -> List[int]:
    n = ctypes.c_size_t()
    buf = ctypes.create_string_buffer(data, len(data))
    _check(N.lib().gkr_wtns_parse(buf, ctypes.c_size_t(len(data)), None, ctypes.c_size_t(0), ctypes.byref(n)), "gkr_wtns_parse")
    out = np.zeros((max(n.value, 1), 4), dtype=np.uint64)
    _check(N.lib().gkr_wtns_parse(buf, ctypes.c_size_t(len(data)), _ptr(out), ctypes.c_size_t(n.value), ctypes.byref(n)),
           "gkr_wtns_parse")
    return from_limbs(out[:n.value])


def write_wtns(values: Sequence[int]) -> bytes:
    vl = to_limbs(list(values)) if len(values) else np.zeros((0, 4), dtype=np.uint64)
    need = ctypes.c_size_t()
    _check(N.lib().gkr_wtns_serialize(_ptr(vl), ctypes.c_size_t(len(values)), None, ctypes.c_size_t(0), ctypes.byref(need)),
           "gkr_wtns_serialize")
    buf = ctypes.create_string_buffer(need.value)
    _check(N.lib().gkr_wtns_serialize(_ptr(vl), ctypes.c_size_t(len(values)), buf, ctypes.c_size_t(need.value), ctypes.byref(need)),
           "gkr_wtns_serialize")
    return buf.raw


def convert_r1cs_wtns_gkr(r1cs: R1cs, witness: Sequence[int]):
    """convert_r1cs_wtns_gkr (convert.rs:667-785): -> (circuits, inputs), circuits[j] a GKRCircuit and inputs[j] the
    values of its input layer for this witness; prove(circuits[j], inputs[j], require_zero_output=True) is the
    reference's prover::prove(&circuit, &input) with calculate_input's assertion (:838)."""
    layered = r1cs.compile()
    try:
        wl = as_limbs(witness)
        circuits = [layered.circuit(j) for j in range(len(layered))]
        inputs = [from_limbs(layered.input_values_raw(j, wl)) for j in range(len(layered))]
    finally:
        layered.close()
    return circuits, inputs
