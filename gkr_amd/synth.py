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
