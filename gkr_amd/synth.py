"""Seeded synthetic workloads of BASELINE.json's configs, shared by bench.py, tools/ and the parity tests so
that what is timed is exactly what is checked (PRNG = numpy PCG64, seed 0xC0FFEE + config).

Everything here is numpy arrays in the C ABI's layout (uint64 limbs (n, 4), uint8 / uint32 gate arrays)."""

import hashlib
import json
import os

import numpy as np

from .prover import GKRCircuit, Layer

SEED = 0xC0FFEE


def rand_fr(rng, count):
    """`count` field elements below 2^253 (< r) as (count, 4) uint64 limbs."""
    a = rng.integers(0, 1 << 63, (count, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 61) - 1)
    return a


def config5_layer(k_i=24, k=12, seed=SEED + 5):
    """configs[4]: one GKR layer with 2^k_i random gates over a 2^k-entry next layer, random z and W."""
    rng = np.random.default_rng(seed)
    g = 1 << k_i
    lay = Layer(k_i, rng.integers(0, 2, g, dtype=np.uint8), rng.integers(0, 1 << k, g, dtype=np.uint32),
                rng.integers(0, 1 << k, g, dtype=np.uint32))
    z, W = rand_fr(rng, k_i), rand_fr(rng, 1 << k)
    return lay, z, W


def circom_shaped_layer(k_i=20, k=20, seed=SEED + 9):
    """A WIDE layer with the structure the reference's compiler emits (rust/src/convert.rs:209-214, 278-343; read off the
    layers gkr_amd.convert compiles from the demo R1CS): gates come in index order of the values they feed on --
      * a quarter: mult gates, every other one reading ONE hot wire (slot 1: a bucket with an eighth of the layer), the rest a
        value and its neighbour, in ascending order;
      * a quarter: add gates over adjacent pairs (2j, 2j + 1);
      * the rest: relay gates Add(x, zero) carrying values up unchanged, x ascending, ALL reading the zero slot 0 as their
        right operand (one bucket with half the layer), the tail padding gates Add(0, 0);
    so a value feeds one gate, runs of gates read runs of values, and two wires feed most of the layer -- unlike
    config5_layer's uniform draw, which has no locality for a kernel to use.  Random z and W.  2^k >= 2^k_i / 2."""
    rng = np.random.default_rng(seed)
    g, m = 1 << k_i, 1 << k
    q = g >> 2
    gt = np.zeros(g, dtype=np.uint8)
    l = np.zeros(g, dtype=np.uint32)
    r = np.zeros(g, dtype=np.uint32)
    j = np.arange(q, dtype=np.uint64)
    gt[:q] = 1
    l[:q] = np.where(j & 1, 1, 2 + j // 2) % m
    r[:q] = np.where(j & 1, 2 + j // 2, 2 + j // 2 + (j // 2) % 2) % m
    base = 2 + q // 2
    l[q:2 * q] = (base + 2 * j) % m
    r[q:2 * q] = (base + 2 * j + 1) % m
    relays = g - 2 * q - (g >> 6)
    jr = np.arange(relays, dtype=np.uint64)
    l[2 * q:2 * q + relays] = (base + 2 * q + jr) % m
    lay = Layer(k_i, gt, l, r)
    z, W = rand_fr(rng, k_i), rand_fr(rng, 1 << k)
    return lay, z, W


def wide_circuit(ks=(18, 20, 20), seed=SEED + 7):
    """A circuit with WIDE layers (2^k[i] random gates over 2^k[i+1] values) and one witness: what bench.py's wide_prove leg
    and tools/bench_wide_prove.py prove.  -> (GKRCircuit, [(gate_type, left, right)], witness limbs (1, 2^k[-1], 4))."""
    rng = np.random.default_rng(seed)
    ks = list(ks)
    raw = []
    for i in range(len(ks) - 1):
        g, m = 1 << ks[i], 1 << ks[i + 1]
        raw.append((rng.integers(0, 2, g, dtype=np.uint8), rng.integers(0, m, g, dtype=np.uint32), rng.integers(0, m, g, dtype=np.uint32)))
    circuit = GKRCircuit([Layer(ks[i], *raw[i]) for i in range(len(raw))], ks[-1])
    return circuit, raw, rand_fr(rng, 1 << ks[-1])[None]


def proof_arrays_digest(ks, sc, sl, sr, q, ql, z, rr):
    """sha256 over one proof's raw arrays (round coefficients | lengths | challenges | q | q lengths | z | r) in the C ABI's
    layout -- the product's gkr_prove_batch outputs for proof 0, or the same arrays assembled from the CPU checker's
    prove_raw (proof_arrays_from_checker)."""
    return transcript_digest(sc, sl, sr, q, ql, z, rr)


def proof_coeffs_digest(d_coeffs, input_coeffs):
    """sha256 over one proof's d and input_func coefficient arrays (get_multi_ext, poly.rs:502-536: all 2^k_0 and 2^k_L monomial
    coefficients, zeros included) in the C ABI's layout."""
    return transcript_digest(np.ascontiguousarray(d_coeffs), np.ascontiguousarray(input_coeffs))


def proof_arrays_from_checker(ref, ks):
    """cdense.prove_raw's dict -> the seven arrays in gkr_proof_buf layout (one proof)."""
    L = len(ks) - 1
    sc = np.concatenate(ref["C"])
    sl = np.concatenate(ref["L"]).astype(np.uint32)
    sr = np.concatenate(ref["R"])
    q = np.concatenate(ref["q"])
    ql = np.asarray(ref["q_len"], dtype=np.uint32)
    z = np.concatenate([x.reshape(-1, 4) for x in ref["z"]]) if sum(ks) else np.zeros((1, 4), dtype=np.uint64)
    rr = np.asarray(ref["r"], dtype=np.uint64).reshape(L, 4)
    return sc, sl, sr, q, ql, z, rr


def bench_table_seed(rank, b):
    """Seed of table b of rank `rank` in bench.py's default workload (1024 x 2^20 points per rank); rank 0's table 0 is
    the table of tests/golden/config_hashes.json["mle"]["n=20,seed=12648432"]."""
    return SEED + 2 + 1000 * rank + b


def bench_batch_digests():
    """tests/golden/bench_batch_hashes.json (make_config_hashes.py --bench-batch): committed digests of the
    reference-semantics transcripts of every table bench.py times, or None."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bench_batch_hashes.json")
    try:
        with open(path) as f:
            return json.load(f)
    except OSError:
        return None


PROOF_BATCH_KS = [5, 6, 7, 7, 7]


def proof_batch_circuit(ks=None, seed=SEED + 3):
    """The synthetic stand-in of configs[3]'s circuit (size class of SURVEY appendix B.4): 4 gate layers,
    k = [5, 6, 7, 7 | input 7], random add / mult gates."""
    ks = list(ks or PROOF_BATCH_KS)
    rng = np.random.default_rng(seed)
    layers = [Layer(ks[i], rng.integers(0, 2, 1 << ks[i], dtype=np.uint8),
                    rng.integers(0, 1 << ks[i + 1], 1 << ks[i], dtype=np.uint32),
                    rng.integers(0, 1 << ks[i + 1], 1 << ks[i], dtype=np.uint32)) for i in range(len(ks) - 1)]
    return GKRCircuit(layers, ks[-1])


def proof_batch_witnesses(n_proofs, input_k=PROOF_BATCH_KS[-1], first=0):
    """witness i = 2^input_k field elements from PCG64(1000 + i); (n_proofs, 2^input_k, 4) uint64."""
    return np.stack([rand_fr(np.random.default_rng(1000 + first + i), 1 << input_k) for i in range(n_proofs)])


def proof_digest(sumcheck_proofs, sumcheck_r, q, z, r):
    """sha256 over one proof's (sumcheck_proofs, sumcheck_r, q, z, r) as nested lists of integers (the reference's
    Proof fields, gkr.rs:7-19, trimmed vectors): brackets mark the nesting, every integer is 32 little-endian bytes.
    Both the library's decoded proofs and the CPU checker's produce these lists."""
    h = hashlib.sha256()

    def put(x):
        if isinstance(x, (list, tuple)):
            h.update(b"[")
            for y in x:
                put(y)
            h.update(b"]")
        else:
            h.update(int(x).to_bytes(32, "little"))
    put([sumcheck_proofs, sumcheck_r, q, z, r])
    return h.hexdigest()


def demo_proof_inputs(n_inputs):
    """The (in1, in2) pairs bench.py proves for configs[3] (input i of `n_inputs`)."""
    return [(2 + i, 3 + (i % 5)) for i in range(n_inputs)]


def proof_digests():
    """tests/golden/proof_digests.json (make_config_hashes.py --proofs): per input and sub-circuit the digest of the CPU
    checker's proof of the demo circuit, for configs[0]'s three example inputs and configs[3]'s 64; or None."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "proof_digests.json")
    try:
        with open(path) as f:
            return json.load(f)
    except OSError:
        return None


def large_r1cs_digests():
    """tests/golden/large_r1cs_digests.json (make_config_hashes.py --large-r1cs): the CPU checker's compile and proofs of the
    262 144-constraint R1CS bench.py's large_r1cs leg proves, per sub-circuit; or None."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "large_r1cs_digests.json")
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def transcript_digest(*arrays):
    """sha256 over the raw output arrays of a sumcheck (coefficients | lengths | challenges)."""
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def golden_digest(kind, key):
    """The committed digest of the reference-semantics transcript of a workload (tests/golden/config_hashes.json,
    written by tests/golden/make_config_hashes.py), or None when that size has none."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "config_hashes.json")
    try:
        with open(path) as f:
            return json.load(f).get(kind, {}).get(key)
    except OSError:
        return None


# ---------------------------------------------------------------------------------------- configs[0] / configs[3]
# rust/t.circom: circomlib MiMC7(91) of the public input in1 with key 0 (in2 is a private input that is not used),
# public output out.  circom is not available in this image, so its R1CS is written by hand: four quadratic
# constraints per round after circom's linear simplification.  Wires: 0 one, 1 out, 2 in1, 3 in2, then t2, t4, t6, t7
# of every round.  circom's own coefficient signs and wire order are NOT reproduced (byte-equality with its
# output is impossible without it); style "negated" writes (-A) * B = -C, the shape circom tends to emit.

def _mimc7_constants():
    import ctypes
    from . import _native as N
    from .field import from_limbs
    out = np.zeros((91, 4), dtype=np.uint64)
    for i in range(91):
        rc = N.lib().gkr_mimc7_constant(ctypes.c_int(i), out[i].ctypes.data_as(ctypes.c_void_p))
        if rc:
            raise RuntimeError("gkr_mimc7_constant(%d) -> %d" % (i, rc))
    return from_limbs(out)


def mimc7_demo_constraints(nrounds=91, style="plain"):
    """-> (n_wires, constraints) with constraints as [(A, B, C)], each a list of (coefficient, wire)."""
    from .field import MODULUS as P
    cts = _mimc7_constants()
    wire, cons, prev_t7 = 4, [], None

    def emit(a, b, c):
        if style == "negated":
            a = [((P - co) % P, w) for co, w in a]
            c = [((P - co) % P, w) for co, w in c]
        cons.append((a, b, c))
    for i in range(nrounds):
        ci = cts[i % len(cts)]    # (more than 91 rounds: the constants repeat -- a long chain for R1CS of any size, not MiMC7 any more)
        t = [(1, 2)] if i == 0 else ([(ci, 0), (1, prev_t7)] if ci else [(1, prev_t7)])
        t2, t4, t6 = wire, wire + 1, wire + 2
        wire += 3
        emit(list(t), list(t), [(1, t2)])
        emit([(1, t2)], [(1, t2)], [(1, t4)])
        emit([(1, t4)], [(1, t2)], [(1, t6)])
        if i < nrounds - 1:
            emit([(1, t6)], list(t), [(1, wire)])
            prev_t7 = wire
            wire += 1
        else:
            emit([(1, t6)], list(t), [(1, 1)])
    return wire, cons


def mimc7_demo_r1cs(nrounds=91, style="plain"):
    from .convert import R1cs
    n_wires, cons = mimc7_demo_constraints(nrounds, style)
    return R1cs.build(n_wires, 1, 1, 1, cons)


def mimc7_demo_witness(in1, in2, nrounds=91):
    """What circom's generated witness calculator would produce for mimc7_demo_r1cs (wire order as above)."""
    from .field import MODULUS as P
    cts = _mimc7_constants()
    w = [1, 0, in1 % P, in2 % P]
    t7 = None
    for i in range(nrounds):
        t = in1 % P if i == 0 else (t7 + cts[i % len(cts)]) % P
        t2 = t * t % P
        t4 = t2 * t2 % P
        t6 = t4 * t2 % P
        w += [t2, t4, t6]
        t7 = t6 * t % P
        if i < nrounds - 1:
            w.append(t7)
    w[1] = t7
    return w


EXAMPLE_INPUTS = [(2, 3), (3, 3), (3, 4)]   # rust/example/input{1,2,3}.json: in1, in2
