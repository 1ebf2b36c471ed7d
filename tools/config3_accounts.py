"""configs[3] (64 inputs x 12 sub-circuits through gkr_prove_many): the library's thread accounts and the hashing pieces'
own figures (gkr_host_accounting: lanes filled per piece, time inside the pass function, time spent posting / waiting),
against the hashing floor measured on this host.   python tools/config3_accounts.py [inputs] [threads] [reps]"""
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
    n_inputs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 14
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    step = ProvingStep(synth.mimc7_demo_r1cs())
    inputs = step.inputs_for(np.stack([as_limbs(synth.mimc7_demo_witness(a, b)) for a, b in synth.demo_proof_inputs(n_inputs)]))
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
        hashed = sum(int(arrs[1].sum()) for arrs in step._prepared["outs"])      # field elements hashed per step
        vectors = sum(int((arrs[1] > 0).sum()) for arrs in step._prepared["outs"])
    own, helped, spin, rest, lent, lent_idle, calls, wake = [float(x) / 1e3 for x in buf[:8]]
    pieces, piece_ms, pass_ms = float(buf[8]), float(buf[9]) / 1e3, float(buf[10]) / 1e3
    hist = [int(buf[10 + n]) for n in range(1, 17)]
    floor_thread_ms = hashed * (lanes16 / 3) / 1e3
    print(json.dumps({
        "inputs": n_inputs, "threads": threads, "step_ms_median": round(statistics.median(each), 3), "accounted_step_ms": round(acc_ms, 3),
        "round_vectors_per_step": vectors, "hashed_elements_per_step": hashed, "us_per_3_element_hash_16_lanes": lanes16, "us_scalar": scalar,
        "floor": {"thread_ms_of_pure_16_lane_hashing": round(floor_thread_ms, 2), "floor_ms_on_these_threads": round(floor_thread_ms / threads, 3)},
        "thread_ms": {"own_pieces": round(own, 2), "others_pieces_while_waiting": round(helped, 2), "spinning": round(spin, 2), "launches_setup": round(rest, 2),
                      "lent_by_threads_without_item": round(lent, 2), "those_idle": round(lent_idle, 2)},
        "pieces": {"count": int(pieces), "transcripts_per_piece_histogram_1_to_16": hist,
                   "mean_transcripts_per_piece": round(sum((n + 1) * c for n, c in enumerate(hist)) / max(1.0, pieces), 2),
                   "thread_ms_inside_the_pass_function": round(pass_ms, 2), "thread_ms_copying_round_vectors_out": round(piece_ms - pass_ms, 2),
                   "thread_ms_posting_looking_waiting": round(own + helped + lent - piece_ms, 2),
                   "pass_function_over_pure_hash_floor": round(pass_ms / floor_thread_ms, 3) if floor_thread_ms else None}}))
    step.close()


if __name__ == "__main__":
    main()
