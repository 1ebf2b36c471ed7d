#!/usr/bin/env python3
"""The fold pass's placement lottery, per scheduling group: one process, the headline workload, per-launch bandwidth of
k_mle_multifold_mfma<5> attributed to the group (4 GiB region of the tables) it read.  The launches of a step go out in
group order, so sample i belongs to group i mod groups.  Usage: python tools/lottery_probe.py [steps]"""
import ctypes
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch  # noqa: F401  (torch's runtime first, as in bench.py)
    torch.cuda.init()
    from gkr_amd import Context, synth
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n, batch, groups = 20, 1024, 8
    count = 1 << n
    ctx = Context(0)
    tables = ctx.alloc(batch * count * 32)
    for b in range(batch):
        ctx.fill_table(ctypes.c_void_p(tables.value + b * count * 32), count, synth.bench_table_seed(0, b))
    ctx.synchronize()
    out = None
    for _ in range(2):
        out = ctx.sumcheck_mle_batch_device(tables, n, batch, out=out)
    ctx.profile(2)
    ctx.profile_reset()
    for _ in range(steps):
        out = ctx.sumcheck_mle_batch_device(tables, n, batch, out=out)
    ctx.profile(False)
    samples = [(ms, by) for ms, by in ctx.profile_samples("mle_multifold") if by > 4e9]
    rates = [by / (ms * 1e-3) / 1e9 for ms, by in samples]
    per_group = [round(statistics.median(rates[g::groups])) for g in range(groups)]
    spread = [round(max(rates[g::groups]) - min(rates[g::groups])) for g in range(groups)]
    ceil = ctx.ceilings(1 << 30)
    print(json.dumps({"va": hex(tables.value), "launches": len(rates), "median": round(statistics.median(rates)), "per_group_median": per_group,
                      "per_group_spread": spread, "copy_GBps": round(ceil["copy_GBps"]), "read_GBps": round(ceil["read_GBps"])}))
    ctx.free(tables)
    ctx.close()


if __name__ == "__main__":
    main()
