#!/usr/bin/env python3
"""Fresh process, the headline workload as bench.py runs it: the fold pass's rate STEP BY STEP (median of a step's eight
launches) over 40 steps -- does a slow process climb (clocks ramping) or stay where it started (a mode drawn at start-up)?"""
import ctypes
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch  # noqa: F401
    torch.cuda.init()
    from gkr_amd import Context, synth
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    n, batch = 20, 1024
    count = 1 << n
    t0 = time.perf_counter()
    with Context(0) as ctx:
        t = ctx.alloc(batch * count * 32)
        for b in range(batch):
            ctx.fill_table(ctypes.c_void_p(t.value + b * count * 32), count, synth.bench_table_seed(0, b))
        ctx.synchronize()
        ctx.profile(2)
        out = None
        at = []
        for _ in range(steps):
            out = ctx.sumcheck_mle_batch_device(t, n, batch, out=out)
            at.append(round(time.perf_counter() - t0, 2))
        r = [by / (ms * 1e-3) / 1e9 for ms, by in ctx.profile_samples("mle_multifold") if by > 4e9]
        per_step = [round(statistics.median(r[i * 8:(i + 1) * 8])) for i in range(len(r) // 8)]
        p0 = [by / (ms * 1e-3) / 1e9 for ms, by in ctx.profile_samples("mle_sub_sums") if by > 4e9]
        print(json.dumps({"seconds_at_step_1_and_last": [at[0], at[-1]], "fold_GBps_by_step": per_step,
                          "pass0_GBps_first_and_last_steps": [round(statistics.median(p0[:8])), round(statistics.median(p0[-8:]))]}), flush=True)


if __name__ == "__main__":
    main()
