"""Where inside an allocation do the fold pass's bandwidth modes live?  Allocates the 8 GiB table buffer several
times and, for every 1 GiB chunk of it (32 tables of 2^20 entries), measures the first fold pass and pass 0 on that
chunk alone.  If the speed were a property of the whole allocation every chunk of one allocation would run alike."""
import ctypes
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_amd import Context  # noqa: E402

n, tables_total, per_chunk = 20, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 32
count = 1 << n
ctx = Context(0)
dummies = []
for trial in range(4):
    tables = ctx.alloc(tables_total * count * 32)
    for b in range(tables_total):
        ctx.fill_table(ctypes.c_void_p(tables.value + b * count * 32), count, 1 + b)
    ctx.synchronize()
    ctx.profile(2)
    # chunks in rotation: between two uses of a chunk the other 7 GiB are swept (no warm translations / caches)
    chunks = tables_total // per_chunk
    cold = [[] for _ in range(chunks)]
    for rep in range(4):
        for c in range(chunks):
            ptr = ctypes.c_void_p(tables.value + c * per_chunk * count * 32)
            ctx.profile_reset()
            ctx.sumcheck_mle_batch_device(ptr, n, per_chunk)
            if rep:
                cold[c] += [by / ms / 1e6 for ms, by in ctx.profile_samples("mle_multifold") if by > 0.4 * per_chunk * 33 * (count >> 5) * 32 and ms > 0]
    rows = []
    for c in range(tables_total // per_chunk):
        ptr = ctypes.c_void_p(tables.value + c * per_chunk * count * 32)
        ctx.sumcheck_mle_batch_device(ptr, n, per_chunk)
        ctx.profile_reset()
        for _ in range(4):
            ctx.sumcheck_mle_batch_device(ptr, n, per_chunk)
        fold = [by / ms / 1e6 for ms, by in ctx.profile_samples("mle_multifold") if by > 0.4 * per_chunk * 33 * (count >> 5) * 32 and ms > 0]
        first = [by / ms / 1e6 for ms, by in ctx.profile_samples("mle_sub_sums") if ms > 0]
        rows.append((round(statistics.median(fold)), round(statistics.median(first))))
    whole = None
    ctx.sumcheck_mle_batch_device(tables, n, tables_total)
    ctx.profile_reset()
    for _ in range(3):
        ctx.sumcheck_mle_batch_device(tables, n, tables_total)
    f = ctx.profile_get("mle_multifold")
    s = ctx.profile_get("mle_sub_sums")
    print(json.dumps({"trial": trial, "va": hex(tables.value), "whole_fold_GBps": round(f["bytes"] / f["total_ms"] / 1e6),
                      "whole_pass0_GBps": round(s["bytes"] / s["total_ms"] / 1e6), "chunk_fold_GBps_repeated": [r[0] for r in rows], "chunk_fold_GBps_in_rotation": [round(statistics.median(x)) for x in cold],
                      "chunk_pass0_GBps": [r[1] for r in rows]}), flush=True)
    ctx.profile(0)
    ctx.free(tables)
    dummies.append(ctx.alloc((trial + 1) * 37 * (1 << 20)))
ctx.close()
