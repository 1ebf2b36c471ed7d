#!/usr/bin/env python3
"""Is the fold pass's bandwidth "mode" a property of WHICH allocation the tables sit in, or of the context's workspaces, or of
the process?  One process: two table allocations A and B (32 GiB each), contexts created and closed in turn; per (allocation,
context) the median rate of k_mle_multifold_mfma<5> over a few steps of the headline workload.
    python tools/lottery_probe2.py [rounds]"""
import ctypes
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rate(ctx, tables, n, batch, steps=4):
    out = None
    for _ in range(2):
        out = ctx.sumcheck_mle_batch_device(tables, n, batch, out=out)
    ctx.profile(2)
    ctx.profile_reset()
    for _ in range(steps):
        out = ctx.sumcheck_mle_batch_device(tables, n, batch, out=out)
    ctx.profile(False)
    r = [by / (ms * 1e-3) / 1e9 for ms, by in ctx.profile_samples("mle_multifold") if by > 4e9]
    return round(statistics.median(r))


def main():
    import torch  # noqa: F401
    torch.cuda.init()
    from gkr_amd import Context, synth
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n, batch = 20, 1024
    count = 1 << n
    owner = Context(0)          # owns the table allocations for the whole run
    allocs = {}
    for name in "AB":
        t = owner.alloc(batch * count * 32)
        for b in range(batch):
            owner.fill_table(ctypes.c_void_p(t.value + b * count * 32), count, synth.bench_table_seed(0, b))
        allocs[name] = t
    owner.synchronize()
    print(json.dumps({k: hex(v.value) for k, v in allocs.items()}))
    for rnd in range(rounds):
        with Context(0) as ctx:          # fresh workspaces, pinned records, streams
            row = {"context": rnd}
            for name in ("A", "B", "A", "B"):
                row.setdefault(name, []).append(rate(ctx, allocs[name], n, batch))
            print(json.dumps(row), flush=True)
    # a third allocation made late (after the contexts' workspaces came and went)
    t = owner.alloc(batch * count * 32)
    for b in range(batch):
        owner.fill_table(ctypes.c_void_p(t.value + b * count * 32), count, synth.bench_table_seed(0, b))
    with Context(0) as ctx:
        print(json.dumps({"late allocation C": hex(t.value), "C": [rate(ctx, t, n, batch), rate(ctx, t, n, batch)], "A": [rate(ctx, allocs["A"], n, batch)]}))
    owner.close()


if __name__ == "__main__":
    main()
