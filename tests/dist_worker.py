"""Worker of the world_size-2 gloo test (CPU only): drives gkr_amd.parallel's distributed
algorithms -- the per-round all-reduce of limb-widened field elements, the all-gather, the
redundant tail -- with shard objects computed by the CPU ORACLE (there is no GPU here; the
product's shard object on a GPU box is the library session).  Launched by
tests/test_distributed_gloo.py through torch.distributed.run.
"""

import json
import os
import random
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import torch.distributed as dist  # noqa: E402

from gkr_amd import parallel  # noqa: E402
from oracle import dense, gatesum  # noqa: E402
from oracle.field import P  # noqa: E402
from oracle.mimc7 import multi_hash as oracle_hash  # noqa: E402


class OracleLayerShard:
    """Pure-Python twin of the library's layer session (sums / bind / tail) for shard p of nshards."""

    def __init__(self, k_i, k, gt, l, r, z, w, nshards, shard):
        lp = nshards.bit_length() - 1
        self.k, self.kc, self.round = k, k - lp, 0
        e = dense.eq_table(z) if k_i else [1]
        n = 1 << (2 * k - lp)
        self.a, self.m = [0] * n, [0] * n
        for g, ty in enumerate(gt):
            if r[g] % nshards != shard:
                continue
            idx = (l[g] << self.kc) | (r[g] >> lp)
            tgt = self.m if ty else self.a
            tgt[idx] = (tgt[idx] + e[g]) % P
        self.wb = [x % P for x in w]
        self.wc = [w[i * nshards + shard] % P for i in range(1 << self.kc)]

    def sums(self):
        h = len(self.a) // 2
        c0 = g1 = c2 = 0
        bphase = self.round < self.k
        hb = h >> self.kc
        for i in range(h):
            a0, a1, m0, m1 = self.a[i], self.a[i + h], self.m[i], self.m[i + h]
            if bphase:
                row, col = i >> self.kc, i & ((1 << self.kc) - 1)
                p0, p1, q0, q1 = self.wb[row], self.wb[row + hb], self.wc[col], self.wc[col]
            else:
                p0 = p1 = self.wb[0]
                q0, q1 = self.wc[i], self.wc[i + h]
            c0 += a0 * (p0 + q0) + m0 * p0 * q0
            g1 += a1 * (p1 + q1) + m1 * p1 * q1
            c2 += (a1 - a0) * ((p1 + q1) - (p0 + q0)) + (m1 - m0) * (p1 * q1 - p0 * q0)
        return [c0 % P, g1 % P, c2 % P]

    def bind(self, r):
        def fold(t):
            h = len(t) // 2
            return [(t[i] + r * (t[i + h] - t[i])) % P for i in range(h)]
        self.a, self.m = fold(self.a), fold(self.m)
        if self.round < self.k:
            self.wb = fold(self.wb)
        else:
            self.wc = fold(self.wc)
        self.round += 1

    def tail(self):
        return [self.a[0], self.m[0], self.wc[0], self.wb[0]]


class OracleLayerTail(OracleLayerShard):
    def __init__(self, kc, A, M, wb, Wc):
        self.k = self.kc = kc
        self.round = kc
        self.a, self.m, self.wb, self.wc = list(A), list(M), [wb], list(Wc)


class OracleMleShard:
    def __init__(self, table):
        self.t = [x % P for x in table]
        n = len(table)
        self.dep = any(self.t[2 * i] != self.t[2 * i + 1] for i in range(n // 2)) if n >= 2 else False

    def sums(self):
        h = len(self.t) // 2
        return [sum(self.t[:h]) % P, sum(self.t[h:]) % P], self.dep

    def bind(self, r):
        h = len(self.t) // 2
        self.t = [(self.t[i] + r * (self.t[i + h] - self.t[i])) % P for i in range(h)]

    def value(self):
        return self.t[0]


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    coll = parallel.TorchCollective()
    out = {"rank": rank, "world": world}

    # independent units: contiguous split, no collective
    out["units"] = list(parallel.shard_units(11, rank, world))

    # collective plumbing: modular all-reduce through widened limbs, all-gather, or
    vals = [(P - 1 - rank) % P, 12345 + rank, 0]
    out["allreduce"] = [str(v) for v in coll.all_reduce_fr(vals)]
    out["allgather"] = [[str(v) for v in row] for row in coll.all_gather_fr([rank + 7])]
    out["or"] = coll.all_reduce_or(rank == world - 1)

    # distributed GKR layer sumcheck (every rank builds the same seeded instance, keeps its shard)
    rng = random.Random(2026)
    k_i, k = 3, 3
    g = 1 << k_i
    gt = [rng.randint(0, 1) for _ in range(g)]
    l = [rng.randrange(1 << k) for _ in range(g)]
    r = [rng.randrange(1 << k) for _ in range(g)]
    z = [rng.randrange(P) for _ in range(k_i)]
    w = [rng.randrange(P) for _ in range(1 << k)]
    shard = OracleLayerShard(k_i, k, gt, l, r, z, w, world, rank)
    proof, rs = parallel.prove_sumcheck_opt_distributed(shard, coll, k, dense.depends_on(w, k),
                                                        lambda kc, A, M, wb, Wc: OracleLayerTail(kc, A, M, wb, Wc),
                                                        hasher=oracle_hash)
    ref = dense.sumcheck_layer(k_i, k, gt, l, r, z, w)
    out["layer_ok"] = (proof, rs) == ref

    # a W that lacks a variable: short round vectors must survive the sharding
    w2 = [(i >> (k - 1)) + 1 for i in range(1 << k)]
    shard = OracleLayerShard(k_i, k, gt, l, r, z, w2, world, rank)
    got = parallel.prove_sumcheck_opt_distributed(shard, coll, k, dense.depends_on(w2, k),
                                                  lambda kc, A, M, wb, Wc: OracleLayerTail(kc, A, M, wb, Wc),
                                                  hasher=oracle_hash)
    out["layer_short_ok"] = got == dense.sumcheck_layer(k_i, k, gt, l, r, z, w2)

    # the gate-sharded form (gkr_sumcheck_layer_sharded): each rank sums its own gate range, the library's
    # sum-over-ranks hook (make_allreduce_hook over this gloo group: widen -> SUM all-reduce -> narrow) completes
    # U, V and then the row -- the hook and the collective are the product's, the per-rank sums the oracle's
    import ctypes
    import numpy as np
    from gkr_amd.field import from_limbs, to_limbs
    hook, hook_errors = parallel.make_allreduce_hook(coll.sum_limbs)

    def reduce_through_hook(parts):   # parts: this rank's vector only
        buf = to_limbs(parts[0])
        assert hook(None, buf.ctypes.data_as(ctypes.c_void_p), len(parts[0])) == 0 and not hook_errors
        return from_limbs(buf)
    k_i2, k2 = 5, 3
    g2 = 1 << k_i2
    gt2 = [rng.randint(0, 1) for _ in range(g2)]
    l2 = [rng.randrange(1 << k2) for _ in range(g2)]
    r2 = [rng.randrange(1 << k2) for _ in range(g2)]
    z2 = [rng.randrange(P) for _ in range(k_i2)]
    ok = True
    for wv in ([rng.randrange(P) for _ in range(1 << k2)], [(i >> (k2 - 1)) + 1 for i in range(1 << k2)]):
        first, count = parallel.gate_range(k_i2, rank, world)
        dep = dense.depends_on(wv, k2)
        uv = reduce_through_hook([sum(gatesum.partial_uv(k_i2, k2, gt2, l2, r2, z2, wv, first, count), [])])
        nn = 1 << k2
        pb, rb, wu = gatesum.rounds_b(uv[:nn], uv[nn:], wv, dep)
        rows = reduce_through_hook([sum(gatesum.partial_rows(k_i2, k2, gt2, l2, r2, z2, rb, first, count), [])])
        pc, rc = gatesum.rounds_c(rows[:nn], rows[nn:], wv, wu, dep)
        ok = ok and (pb + pc, rb + rc) == dense.sumcheck_layer(k_i2, k2, gt2, l2, r2, z2, wv)
    out["gate_sharded_ok"] = ok

    # distributed plain sumcheck
    n = 6
    table = [rng.randrange(P) for _ in range(1 << n)]
    local = OracleMleShard(table[rank::world])
    dep_rank_bit = any(table[2 * i] != table[2 * i + 1] for i in range(len(table) // 2))
    got = parallel.prove_sumcheck_distributed(local, coll, n, dep_rank_bit, lambda v: OracleMleShard(v), hasher=oracle_hash)
    out["mle_ok"] = got == dense.sumcheck_mle(table, n)

    # the split of gkr_sumcheck_mle_sharded_dev (rank = index bits log2 P .. 1, the last variable inside every shard,
    # sums linear over ranks, gathered tail) restated on Python integers over this gloo group, against the oracle:
    # tables with and without rank-local rounds, one that does not depend on its last variable, one constant
    ok = True
    lp = world.bit_length() - 1
    for n2, kind in ((9, "random"), (9, "no-last"), (8, "constant"), (5, "random"), (7, "no-last")):
        if kind == "random":
            tb = [rng.randrange(P) for _ in range(1 << n2)]
        elif kind == "no-last":
            half = [rng.randrange(P) for _ in range(1 << (n2 - 1))]
            tb = [half[i >> 1] for i in range(1 << n2)]
        else:
            tb = [11] * (1 << n2)
        mine = [tb[h * 2 * world + 2 * rank + x] for h in range(1 << (n2 - lp - 1)) for x in (0, 1)]
        ok = ok and parallel.prove_sumcheck_split_model(mine, n2, lp, rank, coll, hasher=oracle_hash) == dense.sumcheck_mle(tb, n2)
    out["mle_split_ok"] = ok

    with open(os.path.join(os.environ["GKR_TEST_OUT"], "rank%d.json" % rank), "w") as f:
        json.dump(out, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
