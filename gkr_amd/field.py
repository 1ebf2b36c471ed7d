"""Field-element marshalling for the C ABI: Python ints <-> (n, 4) uint64 limbs.

bn256::Fr crosses the boundary as its 32-byte little-endian canonical repr
(rust/src/gkr/sumcheck.rs:10-22; decimal strings only at the JSON edge,
rust/src/file_utils.rs:20-28).
"""

import numpy as np

MODULUS = 21888242871839275222246405745257275088548364400416034343698204186575808495617
_M64 = (1 << 64) - 1


def to_limbs(values):
    """Iterable of ints (any residue) -> canonical limbs, shape (n, 4) uint64."""
    values = list(values)
    out = np.empty((len(values), 4), dtype=np.uint64)
    for i, v in enumerate(values):
        v %= MODULUS
        out[i, 0] = v & _M64
        out[i, 1] = (v >> 64) & _M64
        out[i, 2] = (v >> 128) & _M64
        out[i, 3] = v >> 192
    return out


def from_limbs(arr):
    a = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [int(a[i, 0]) | (int(a[i, 1]) << 64) | (int(a[i, 2]) << 128) | (int(a[i, 3]) << 192)
            for i in range(a.shape[0])]


def as_limbs(x):
    """Accept an (n, 4) uint64 array (passed through) or an iterable of ints."""
    if isinstance(x, np.ndarray) and x.dtype == np.uint64 and x.ndim == 2 and x.shape[1] == 4:
        return np.ascontiguousarray(x)
    return to_limbs(x)
