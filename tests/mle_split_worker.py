"""Worker of tests/test_gpu_mle_split.py::test_split_sumcheck_two_processes_over_gloo_one_gpu: one rank of a
torch.distributed world (backend gloo; the ranks share the one visible GPU), its shard of a seeded table through
gkr_sumcheck_mle_sharded_dev with the staged device exchange."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    torch.cuda.init()
    torch.cuda.set_device(0)
    from gkr_amd import Context, parallel, synth
    from oracle import cdense
    rank, world = dist.get_rank(), dist.get_world_size()
    lp = world.bit_length() - 1
    coll = parallel.TorchCollective()
    ok, exchanges, digest = True, [], []
    with Context(0) as ctx:
        for n, batch in ((16, 1), (20, 2), (4, 1)):
            tables = np.stack([cdense.fill_table(1 << n, 31 * n + b) for b in range(batch)])
            mine = np.stack([parallel.mle_shard(tables[b], n, lp, rank) for b in range(batch)])
            ex = coll.device_exchange(parallel.exchange_limbs_mle(n, lp, batch))
            d = ctx.alloc(mine.nbytes)
            ctx.upload(d, mine)
            C, L, R, nx = parallel.sumcheck_mle_sharded_raw(ctx, d, n, lp, rank, ex, batch)
            ctx.free(d)
            exchanges.append(nx)
            digest.append(synth.transcript_digest(C, L, R))
            for b in range(batch):
                want = cdense.sumcheck_mle_raw(tables[b], n)
                ok = ok and np.array_equal(C[b], want[0]) and np.array_equal(L[b], want[1]) and np.array_equal(R[b], want[2])
    with open(os.path.join(os.environ["GKR_TEST_OUT"], "split_rank%d.json" % rank), "w") as f:
        json.dump({"ok": bool(ok), "world": world, "backend": dist.get_backend(), "exchanges": exchanges, "digest": digest}, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
