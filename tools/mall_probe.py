"""Can the fold pass read its tables from the 256 MiB Infinity Cache instead of HBM?

The default schedule reads every input table twice from HBM: pass 0 (sub-block sums) over a whole group of tables,
then -- after the host has hashed -- the first fold pass over the same group; a group is 8 GiB, so the second read
never hits the memory-side cache.  This probe proves sumchecks in SMALL groups (B tables of 2^20 entries = B x 32
MiB per call), cycling over enough distinct table sets that pass 0 always reads HBM-cold data, and prints the
bandwidth of pass 0 and of the first fold pass per group size: when B x 32 MiB (plus what moves in between) fits
the cache, the fold pass finds its source there.

    python tools/mall_probe.py            # B = 1, 2, 3, 4, 6, 8, 16, 64
"""
import ctypes
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_amd import Context  # noqa: E402

n = 20
count = 1 << n
total_tables = 256   # 8 GiB: every set is long evicted when its turn comes again
ctx = Context(0)
tables = ctx.alloc(total_tables * count * 32)
for b in range(total_tables):
    ctx.fill_table(ctypes.c_void_p(tables.value + b * count * 32), count, 1 + b)
ctx.synchronize()
out = []
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6, 8, 16, 64]:
    sets = total_tables // B
    ctx.profile(1)
    for s in range(min(sets, 4)):   # warm-up: workspaces, code objects
        ctx.sumcheck_mle_batch_device(ctypes.c_void_p(tables.value + s * B * count * 32), n, B)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for s in range(sets):
        ctx.sumcheck_mle_batch_device(ctypes.c_void_p(tables.value + s * B * count * 32), n, B)
    wall = time.perf_counter() - t0
    first = ctx.profile_samples("mle_sub_sums")
    fold = [(ms, by) for ms, by in ctx.profile_samples("mle_multifold") if by > 0.9 * B * 33 * (count >> 5) * 32]
    rate = lambda xs: statistics.median(by / (ms * 1e-3) / 1e9 for ms, by in xs if ms > 0) if xs else None
    row = {"tables_per_call": B, "MiB_per_call": B * 32, "calls": sets, "pass0_GBps_median": rate(first), "first_fold_GBps_median": rate(fold),
           "pass0_us_median": statistics.median(ms * 1e3 for ms, _ in first) if first else None,
           "first_fold_us_median": statistics.median(ms * 1e3 for ms, _ in fold) if fold else None,
           "wall_ms_per_call": wall / sets * 1e3}
    out.append(row)
    print(json.dumps(row), flush=True)
    ctx.profile(0)
ctx.free(tables)
ctx.close()
