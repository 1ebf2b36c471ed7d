import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for t in (1, 2, 4, 8, 12):
    os.environ["GKR_BENCH_PROOF_THREADS"] = str(t)
    r = bench.proofs_per_sec(0, 64)
    print(t, "threads:", round(r["proofs_per_sec"]), "proofs/s", round(r["ms_per_proof_per_thread"], 2), "ms per proof per thread", flush=True)
