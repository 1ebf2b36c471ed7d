"""Child process of tests/test_gpu_parity.py::test_mle_largest_tables: one plain sumcheck on a 2^n-point table generated on
the device (n up to the ABI's 30: a 32 GiB table).  n <= 28: every byte of the transcript against the C oracle on the
same table; above: the size-independent relations g_j(0) + g_j(1) = g_{j-1}(r_{j-1}), r_j = MiMC7(g_j), plus the first
round against the table's sum, which the device table's own second run (a fresh output) must reproduce."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_amd import Context, multi_hash  # noqa: E402
from gkr_amd.field import MODULUS as P, from_limbs  # noqa: E402


def main():
    n, seed = int(sys.argv[1]), 0xB16 + int(sys.argv[1])
    count = 1 << n
    with Context(0) as ctx:
        d = ctx.alloc(count * 32)
        try:
            ctx.fill_table(d, count, seed)
            C, L, R = ctx.sumcheck_mle_batch_device(d, n, 1)
            C2, L2, R2 = ctx.sumcheck_mle_batch_device(d, n, 1)
        finally:
            ctx.free(d)
    if not (np.array_equal(C, C2) and np.array_equal(L, L2) and np.array_equal(R, R2)):
        print("MISMATCH between two runs")
        return 1
    claim = None
    rs = from_limbs(R[0])
    for j in range(n):
        g = from_limbs(C[0, j])[2 - int(L[0, j]):]
        if claim is not None and (g[-1] + sum(g)) % P != claim:
            print("MISMATCH sum relation, round", j)
            return 1
        if multi_hash(g) != rs[j]:
            print("MISMATCH challenge, round", j)
            return 1
        claim = 0
        for c in g:
            claim = (claim * rs[j] + c) % P
    if n <= 28:
        from oracle import cdense
        want = cdense.sumcheck_mle_raw(cdense.fill_table(count, seed), n)
        if not (np.array_equal(C[0], want[0]) and np.array_equal(L[0], want[1]) and np.array_equal(R[0], want[2])):
            print("MISMATCH against the oracle")
            return 1
    # the table bench.py --mode mle-split splits over the ranks (seed SEED + 2): its transcript against the committed digest
    # of the CPU checker's (tests/golden/config_hashes.json; n = 30 made with the in-place C prover, 32 GiB of host memory)
    from gkr_amd import synth
    want = synth.golden_digest("mle", "n=%d,seed=%d" % (n, synth.SEED + 2))
    if want is not None:
        with Context(0) as ctx:
            d = ctx.alloc(count * 32)
            try:
                ctx.fill_table(d, count, synth.SEED + 2)
                C, L, R = ctx.sumcheck_mle_batch_device(d, n, 1)
            finally:
                ctx.free(d)
        if synth.transcript_digest(C[0], L[0], R[0]) != want:
            print("MISMATCH against the committed digest of the oracle's transcript")
            return 1
        print("digest of the oracle's 2^%d transcript: equal" % n)
    print("OK")
    return 0


if __name__ == "__main__":
    sys.exit(main())
