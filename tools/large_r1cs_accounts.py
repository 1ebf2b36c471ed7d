"""The 262 144-constraint step (16 sub-circuits, one witness each, gkr_prove_many in lockstep groups): the library's thread accounts
and the hashing pieces' own figures (gkr_host_accounting) -- how long a one-proof piece takes, how much of a pass the group's
thread spends waiting for the pieces others took.   python tools/large_r1cs_accounts.py [threads] [reps]"""
import ctypes
import json
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_amd import Context, synth  # noqa: E402
from gkr_amd import _native as N  # noqa: E402
from gkr_amd.aggregate import ProvingStep  # noqa: E402
from gkr_amd.field import as_limbs  # noqa: E402
from gkr_amd.prover import host_hash_us  # noqa: E402


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    nrounds = 65536
    step = ProvingStep(synth.mimc7_demo_r1cs(nrounds=nrounds))
    inputs = step.inputs_for(np.stack([as_limbs(synth.mimc7_demo_witness(2, 3, nrounds=nrounds))]))
    lib = N.lib()
    lanes16, scalar = host_hash_us(3)
    with Context(0) as ctx:
        for _ in range(4):
            step.prove_raw_many(ctx, inputs, threads)
        each = []
        for _ in range(reps):
            t = time.perf_counter()
            step.prove_raw_many(ctx, inputs, threads)
            each.append((time.perf_counter() - t) * 1e3)
        lib.gkr_host_accounting(1)
        t = time.perf_counter()
        step.prove_raw_many(ctx, inputs, threads)
        acc_ms = (time.perf_counter() - t) * 1e3
        lib.gkr_host_accounting(0)
        buf = (ctypes.c_double * 28)()
        lib.gkr_host_accounting_read(buf, 28)
        vectors = sum(int((arrs[1] > 0).sum()) for arrs in step._prepared["outs"])
    own, helped, spin, rest, lent, lent_idle, calls, wake = [float(x) / 1e3 for x in buf[:8]]
    pieces, piece_ms, pass_ms = float(buf[8]), float(buf[9]) / 1e3, float(buf[10]) / 1e3
    hist = [int(buf[10 + n]) for n in range(1, 17)]
    print(json.dumps({
        "threads": threads, "step_ms_median": round(statistics.median(each), 3), "accounted_step_ms": round(acc_ms, 3),
        "round_vectors_per_step": vectors, "us_per_3_element_hash_16_lanes": lanes16, "us_scalar": scalar,
        "thread_ms": {"own_pieces_incl_waiting_for_helpers": round(own, 2), "others_pieces_while_waiting": round(helped, 2), "spinning": round(spin, 2),
                      "launches_setup": round(rest, 2), "lent_by_threads_without_item": round(lent, 2), "those_idle": round(lent_idle, 2)},
        "pieces": {"count": int(pieces), "transcripts_per_piece_histogram_1_to_16": hist,
                   "thread_ms_inside_the_pass_function": round(pass_ms, 2),
                   "us_inside_the_pass_function_per_piece": round(pass_ms * 1e3 / max(1.0, pieces), 1),
                   "us_per_piece_all": round(piece_ms * 1e3 / max(1.0, pieces), 1)}}))
    step.close()


if __name__ == "__main__":
    main()
