"""Child process of test_gpu_parity.test_fold_pass_variants_match_oracle: the schedule knobs are read once per
process, so every variant runs in its own interpreter.  Usage: python fold_variants_worker.py <n> <batch>"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_amd import Context  # noqa: E402
from oracle import cdense  # noqa: E402


def special_tables(count):
    """Two tables on the fold pass's sign / carry boundaries: all p - 1, and a mix of byte-pattern extremes."""
    from oracle.field import P
    specials = [0, 1, P - 1, P - 2, (1 << 253) - 1, int.from_bytes(b"\x80" * 31 + b"\x20", "little"),
                int.from_bytes(b"\x7f" * 31 + b"\x2f", "little"), int.from_bytes(b"\xff" * 31 + b"\x2f", "little"), 0x80, 0xff]
    limbs = np.array([[(v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)] for v in specials], dtype=np.uint64)
    rng = np.random.default_rng(count)
    return [np.repeat(limbs[2:3], count, axis=0), limbs[rng.integers(0, len(specials), count)]]


def main():
    n, batch = int(sys.argv[1]), int(sys.argv[2])
    count = 1 << n
    tables = [cdense.fill_table(count, 9000 + 31 * n + b) for b in range(batch)]
    tables[:2] = special_tables(count)
    with Context(0) as ctx:
        d = ctx.alloc(batch * count * 32)
        try:
            for b in range(batch):
                if b < 2:
                    ctx.upload(ctypes.c_void_p(d.value + b * count * 32), np.ascontiguousarray(tables[b]))
                else:
                    ctx.fill_table(ctypes.c_void_p(d.value + b * count * 32), count, 9000 + 31 * n + b)
            C, L, R = ctx.sumcheck_mle_batch_device(d, n, batch)
        finally:
            ctx.free(d)
    for b in range(batch):
        c2, l2, r2 = cdense.sumcheck_mle_raw(np.ascontiguousarray(tables[b]), n)
        if not (np.array_equal(C[b], c2) and np.array_equal(L[b], l2) and np.array_equal(R[b], r2)):
            print("MISMATCH sumcheck", b)
            return 1
    print("OK", n, batch)
    return 0


if __name__ == "__main__":
    sys.exit(main())
