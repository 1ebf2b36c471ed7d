"""gkr_prove_many step times by thread count, with the cgroup's CPU throttling counters around each run (a crew that spins
on more threads than the quota allows gets the whole process throttled for the rest of a scheduler period).
    python tools/proof_many_probe.py [inputs]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_amd import Context, synth  # noqa: E402
from gkr_amd.aggregate import ProvingStep  # noqa: E402
from gkr_amd.field import as_limbs  # noqa: E402


def cpu_stat():
    out = {}
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            for line in open(path):
                k, v = line.split()
                out[k] = int(v)
            break
        except OSError:
            pass
    return out


n_inputs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
step = ProvingStep(synth.mimc7_demo_r1cs())
inputs = step.inputs_for(np.stack([as_limbs(synth.mimc7_demo_witness(2 + i, 3 + (i % 5))) for i in range(n_inputs)]))
try:
    print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip(), " affinity:", len(os.sched_getaffinity(0)))
except OSError:
    pass
for threads in [int(x) for x in os.environ.get('PROBE_THREADS', '12,10,8,6,12').split(',')]:
    with Context(0) as ctx:
        for _ in range(3):
            step.prove_raw_many(ctx, inputs, threads)
        before = cpu_stat()
        each = []
        for _ in range(int(os.environ.get('PROBE_REPS', '20'))):
            t = time.perf_counter()
            step.prove_raw_many(ctx, inputs, threads)
            each.append(round((time.perf_counter() - t) * 1e3, 2))
        after = cpu_stat()
        print(json.dumps({"threads": threads, "ms_median": sorted(each)[len(each) // 2], "ms_max": max(each), "over_14ms": sum(1 for x in each if x > 14.0), "ms_each": each if len(each) <= 20 else None,
                          "throttled_periods": after.get("nr_throttled", 0) - before.get("nr_throttled", 0),
                          "throttled_us": after.get("throttled_usec", after.get("throttled_time", 0)) - before.get("throttled_usec", before.get("throttled_time", 0))}), flush=True)
