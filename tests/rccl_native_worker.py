"""Child process of tests/test_gpu_sharded.py::test_native_rccl_exchange_single_rank: the sum over ranks through the
LIBRARY's own RCCL communicator (csrc/exchange_rccl.cpp) -- no torch anywhere in this process.  One rank (one MI355X is
what the box has): RCCL executes every all-reduce of the gate-sharded layer sumcheck and of the split plain sumcheck."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from gkr_amd import Context, parallel, synth
    from oracle import cdense
    uid = parallel.RcclExchange.unique_id()
    assert len(uid) == 128
    with Context(0) as ctx:
        # gate-sharded layer sumcheck, two exchanges per sumcheck
        for k_i, k in ((10, 6), (16, 8), (15, 14)):
            lay, z, W = synth.config5_layer(k_i, k, seed=4000 + k)
            gt, l, r = lay.arrays()
            ex = parallel.RcclExchange(0, uid if k == 6 else parallel.RcclExchange.unique_id(), 0, 1, int(parallel.N.lib().gkr_exchange_limbs(k)))
            gates = parallel.ResidentGates(ctx, k_i, 0, gt, l, r)
            want = cdense.sumcheck_layer_lin_raw(k_i, k, gt, l, r, z, W)
            for _ in range(2):
                got = gates.sumcheck_raw(k, z, W, ex)
                assert all(np.array_equal(a, b) for a, b in zip(got, want)), (k_i, k)
            assert ex.calls == 4, ex.calls
            gates.close()
            ex.close()
        # the plain sumcheck split over ranks: one exchange per pass + the gather
        for n, batch in ((20, 1), (13, 3)):
            tables = np.stack([cdense.fill_table(1 << n, 77 + 5 * b + n) for b in range(batch)])
            ex = parallel.RcclExchange(0, parallel.RcclExchange.unique_id(), 0, 1, parallel.exchange_limbs_mle(n, 0, batch))
            d = ctx.alloc(tables.nbytes)
            ctx.upload(d, tables)
            C, L, R, nx = parallel.sumcheck_mle_sharded_raw(ctx, d, n, 0, 0, ex, batch)
            ctx.free(d)
            assert ex.calls == nx and nx >= 2, (ex.calls, nx)
            for b in range(batch):
                want = cdense.sumcheck_mle_raw(tables[b], n)
                assert np.array_equal(C[b], want[0]) and np.array_equal(L[b], want[1]) and np.array_equal(R[b], want[2]), (n, b)
            ex.close()
    assert "torch" not in sys.modules
    print("OK")


if __name__ == "__main__":
    main()
