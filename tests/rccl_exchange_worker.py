"""Child process of test_device_exchange_over_rccl_single_rank: process group (backend nccl = RCCL) first, GPU second;
a gate-sharded layer sumcheck whose two exchanges run as torch.distributed all-reduces on the library's own stream."""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    torch.cuda.set_device(0)
    from gkr_amd import Context, parallel
    from gkr_amd.field import to_limbs
    from oracle import cdense
    from oracle.field import P
    rng = random.Random(77)
    coll = parallel.TorchCollective()
    ex = coll.exchange()
    assert isinstance(ex, parallel.DeviceExchange)
    with Context(0) as ctx:
        for k_i, k in ((10, 6), (16, 8)):
            g = 1 << k_i
            gt = np.array([rng.randint(0, 1) for _ in range(g)], dtype=np.uint8)
            l = np.array([rng.randrange(1 << k) for _ in range(g)], dtype=np.uint32)
            r = np.array([rng.randrange(1 << k) for _ in range(g)], dtype=np.uint32)
            gates = parallel.ResidentGates(ctx, k_i, 0, gt, l, r)
            for _ in range(3):
                z = to_limbs([rng.randrange(P) for _ in range(k_i)])
                w = to_limbs([rng.randrange(P) for _ in range(1 << k)])
                C, L, R = gates.sumcheck_raw(k, z, w, ex)
                c2, l2, r2 = cdense.sumcheck_layer_raw(k_i, k, gt, l, r, z, w)
                assert np.array_equal(C, c2) and np.array_equal(L, l2) and np.array_equal(R, r2), (k_i, k)
            gates.close()
    assert ex.calls == 12, ex.calls
    # the plain sumcheck's trailing-variable form: one all-reduce of the round's two sums per round through RCCL
    # (gkr_mle_session_* shards, parallel.prove_sumcheck_distributed), against the oracle's transcript
    with Context(0) as ctx:
        n = 12
        table = cdense.fill_table(1 << n, 4321)
        d = ctx.alloc((1 << n) * 32)
        ctx.upload(d, table)
        shard = parallel.MleSession(ctx, d, n)
        proof, rs = parallel.prove_sumcheck_distributed(shard, coll, n, None, None)
        shard.close()
        ctx.free(d)
        from gkr_amd.field import from_limbs
        assert (proof, rs) == cdense.sumcheck_mle(from_limbs(table), n)
    # the plain sumcheck split over ranks on the multi-round schedule (gkr_sumcheck_mle_sharded_dev): its per-pass
    # all-reduces and the gather queued on the library's stream through RCCL; one rank = the whole table
    with Context(0) as ctx:
        for n, batch in ((20, 1), (14, 3), (5, 2)):
            tables = np.stack([cdense.fill_table(1 << n, 555 + 7 * b + n) for b in range(batch)])
            ex2 = coll.device_exchange(parallel.exchange_limbs_mle(n, 0, batch))
            assert isinstance(ex2, parallel.DeviceExchange) and ex2.backend == "nccl"
            d = ctx.alloc(tables.nbytes)
            ctx.upload(d, tables)
            C, L, R, nx = parallel.sumcheck_mle_sharded_raw(ctx, d, n, 0, 0, ex2, batch)
            ctx.free(d)
            assert ex2.calls == nx and (nx == 5 if n == 20 else nx >= 1), (n, nx)   # 4 passes (5 + 3 + 5 + 1 rounds) + the gather
            for b in range(batch):
                want = cdense.sumcheck_mle_raw(tables[b], n)
                assert np.array_equal(C[b], want[0]) and np.array_equal(L[b], want[1]) and np.array_equal(R[b], want[2]), (n, b)
    dist.destroy_process_group()
    print("OK")


if __name__ == "__main__":
    main()
