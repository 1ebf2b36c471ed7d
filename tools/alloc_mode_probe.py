"""Does the fold pass's bandwidth depend on where hipMalloc places the tables?  Re-allocates the table buffer
several times inside one process (dummy allocations in between move it) and prints the fold-pass and first-pass
bandwidth of each placement.  Finding on MI355X / ROCm 7.2: the speed is a property of the allocation (not of the
offset inside it, not of the virtual address): re-allocations land in one of three modes, 5.4 / 5.8 / 6.1 TB/s."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_amd import Context  # noqa: E402

n, batch = 20, int(sys.argv[1]) if len(sys.argv) > 1 else 256
count = 1 << n
ctx = Context(0)
dummies = []
for trial in range(8):
    tables = ctx.alloc(batch * count * 32)
    for b in range(batch):
        ctx.fill_table(ctypes.c_void_p(tables.value + b * count * 32), count, 1 + b)
    ctx.synchronize()
    ctx.profile(2)
    ctx.sumcheck_mle_batch_device(tables, n, batch)
    ctx.profile_reset()
    for _ in range(3):
        ctx.sumcheck_mle_batch_device(tables, n, batch)
    f = ctx.profile_get("mle_multifold")
    s = ctx.profile_get("mle_sub_sums")
    print("trial %d  tables @ 0x%x  fold %.0f GB/s  first pass %.0f GB/s" % (
        trial, tables.value, f["bytes"] / f["total_ms"] / 1e6, s["bytes"] / s["total_ms"] / 1e6), flush=True)
    ctx.profile(0)
    ctx.free(tables)
    if trial % 2 == 1:
        dummies.append(ctx.alloc((trial + 1) * 37 * (1 << 20)))   # shift the next placement
ctx.close()
