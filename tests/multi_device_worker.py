"""Child processes of tests/test_gpu_multi_device.py.

    rccl-rank <uid file> <rank> <world> <device>   one rank of the library-owned RCCL exchange (csrc/exchange_rccl.cpp): the
                                                   gate-sharded layer sumcheck and the split plain sumcheck, this rank's shard,
                                                   against the C checker; rank 0 writes the 128-byte id to the file first
    rccl-alone                                     rank 0 of a world of TWO whose peer never comes: prints CREATING and then
                                                   sits inside gkr_exchange_rccl_create (the parent ends it)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rccl_rank(path, rank, world, device):
    from gkr_amd import Context, parallel, synth
    from oracle import cdense
    if rank == 0:
        uid = parallel.RcclExchange.unique_id()
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(path + ".tmp", path)
    else:
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > 120:
                raise SystemExit("no id from rank 0")
            time.sleep(0.05)
        uid = open(path, "rb").read()
    lp = world.bit_length() - 1
    assert 1 << lp == world
    with Context(device) as ctx:
        k_i, k = 16, 8
        n, batch = 20, 2
        limbs = max(int(parallel.N.lib().gkr_exchange_limbs(k)), parallel.exchange_limbs_mle(n, lp, batch))
        ex = parallel.RcclExchange(device, uid, rank, world, limbs)      # blocks until every rank is here
        # one GKR layer split by gates over the ranks: two all-reduces per sumcheck, every rank ends with the whole transcript
        lay, z, W = synth.config5_layer(k_i, k, seed=5100)
        gt, l, r = lay.arrays()
        first, count = parallel.gate_range(k_i, rank, world)
        gates = parallel.ResidentGates(ctx, k_i, first, gt[first:first + count], l[first:first + count], r[first:first + count])
        want = cdense.sumcheck_layer_lin_raw(k_i, k, gt, l, r, z, W)
        for _ in range(2):
            got = gates.sumcheck_raw(k, z, W, ex)
            assert all(np.array_equal(a, b) for a, b in zip(got, want)), "layer transcript, rank %d" % rank
        gates.close()
        # one plain sumcheck split over the ranks: one all-reduce per pass + the gather
        tables = [cdense.fill_table(1 << n, 600 + b) for b in range(batch)]
        shards = np.stack([parallel.mle_shard(t, n, lp, rank) for t in tables])
        d = ctx.alloc(shards.nbytes)
        ctx.upload(d, shards)
        C, L, R, nx = parallel.sumcheck_mle_sharded_raw(ctx, d, n, lp, rank, ex, batch)
        ctx.free(d)
        for b in range(batch):
            w = cdense.sumcheck_mle_raw(tables[b], n)
            assert np.array_equal(C[b], w[0]) and np.array_equal(L[b], w[1]) and np.array_equal(R[b], w[2]), "mle transcript, rank %d" % rank
        assert ex.calls >= 4 + nx
        ex.close()
    print("OK rank %d" % rank)


def rccl_alone():
    from gkr_amd import parallel
    uid = parallel.RcclExchange.unique_id()
    print("CREATING", flush=True)
    parallel.RcclExchange(0, uid, 0, 2, 1024)      # the second rank never comes
    print("RETURNED", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "rccl-rank":
        rccl_rank(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
    elif sys.argv[1] == "rccl-alone":
        rccl_alone()
