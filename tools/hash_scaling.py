"""The host's sixteen-lane MiMC7 hash on 1 .. 16 threads at once: does a thread keep its single-thread rate when the others hash too?
(The hashing floor of bench.py's host-bound legs multiplies the ONE-thread rate by the threads.)   python tools/hash_scaling.py"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_amd.prover import host_hash_us  # noqa: E402

print("usable cpus:", len(os.sched_getaffinity(0)), open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "")
host_hash_us(3)
for T in (1, 2, 4, 8, 12, 14, 16):
    res = [None] * T

    def work(i):
        vals = [host_hash_us(3)[0] for _ in range(6)]
        res[i] = sorted(vals)[len(vals) // 2]
    th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    print("threads %2d: us per 3-element hash (16 lanes), per thread: min %.3f  mean %.3f  max %.3f   (wall %.2f s)"
          % (T, min(res), sum(res) / T, max(res), time.perf_counter() - t0))
