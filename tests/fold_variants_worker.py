"""Child process of test_gpu_parity.test_fold_pass_variants_match_oracle: the schedule knobs are read once per
process, so every variant runs in its own interpreter.  Usage: python fold_variants_worker.py <n> <batch>"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_amd import Context  # noqa: E402
from oracle import cdense  # noqa: E402


def main():
    n, batch = int(sys.argv[1]), int(sys.argv[2])
    count = 1 << n
    with Context(0) as ctx:
        d = ctx.alloc(batch * count * 32)
        try:
            for b in range(batch):
                ctx.fill_table(ctypes.c_void_p(d.value + b * count * 32), count, 9000 + 31 * n + b)
            C, L, R = ctx.sumcheck_mle_batch_device(d, n, batch)
        finally:
            ctx.free(d)
    for b in range(batch):
        c2, l2, r2 = cdense.sumcheck_mle_raw(cdense.fill_table(count, 9000 + 31 * n + b), n)
        if not (np.array_equal(C[b], c2) and np.array_equal(L[b], l2) and np.array_equal(R[b], r2)):
            print("MISMATCH sumcheck", b)
            return 1
    print("OK", n, batch)
    return 0


if __name__ == "__main__":
    sys.exit(main())
