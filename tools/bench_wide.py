"""Times the layer sumcheck on wide layers (gkr_sumcheck_layer on resident gates) -- what each part costs per width.
usage: python tools/bench_wide.py [k_i,k ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_amd import Context, parallel, synth  # noqa: E402

KERNELS = ["gate_lists", "eq_table_z", "gate_uv", "gate_rows", "layer_prod_pass", "exchange"]


def main():
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(20, 15), (22, 16), (24, 18), (20, 20), (22, 22)]
    with Context(0) as ctx:
        for k_i, k in shapes:
            lay, z, W = (synth.circom_shaped_layer(k_i, k) if os.environ.get("WIDE_SHAPE") == "circom" else synth.config5_layer(k_i, k, seed=1234 + k_i * 100 + k))
            t0 = time.perf_counter()
            ctx.sumcheck_layer_raw(lay, k, z, W)
            t_first = time.perf_counter() - t0
            gt, l, r = lay.arrays()
            res = parallel.ResidentGates(ctx, k_i, 0, gt, l, r)
            times = []
            ctx.profile(True)
            ctx.profile_reset()
            for _ in range(5):
                t0 = time.perf_counter()
                res.sumcheck_raw(k, z, W)
                times.append(time.perf_counter() - t0)
            prof = {name: ctx.profile_get(name) for name in KERNELS}
            ctx.profile(False)
            line = {"k_i": k_i, "k": k, "one_shot_ms": round(t_first * 1e3, 3), "resident_ms": [round(t * 1e3, 3) for t in times],
                    "kernel_ms_per_call": {n: round(p["total_ms"] / max(1, len(times)), 4) for n, p in prof.items() if p["launches"]}}
            print(line, flush=True)
            res.close()


if __name__ == "__main__":
    main()
