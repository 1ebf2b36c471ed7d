#!/usr/bin/env python3
"""Headline benchmark: BN254-Fr sumcheck field-ops/sec on 2^20-point multilinear
tables (BASELINE.json metric, configs[2]), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch: `--batch` (default 1024, 32 GiB of tables) independent
2^n-point multilinear sumchecks (prove_sumcheck, rust/src/gkr/sumcheck.rs:158-214),
tables resident in HBM before the timed region, MiMC7 transcript included.  (One
aggregation step of the reference proves up to 20 sub-circuits with several layers
each, every layer one sumcheck -- aggregator.rs:350-355 -- and BASELINE configs[3]
aggregates 64 inputs: hundreds of independent sumchecks per step are the normal load.)
Independent sumchecks shard across ranks with no data-path collective (weak
scaling: every rank proves its own batch).

field-ops: 5 (2^n - 1) per sumcheck; algorithmic bytes 128 * 2^n per sumcheck
(SURVEY.md section 8d).  The JSON line also carries
  roofline      the dominant kernel (k_mle_multifold_mfma, the fold pass) timed with HIP events on
                the library's stream during the timed steps (profile level 2: only the bandwidth-
                bound kernels carry events, the small round-trip kernels are left alone)
  cpu_baseline  the plain-C oracle (oracle/c, OpenMP over the host cores) on a
                bounded sample of the same workload; rank 0, N = 1 only
"""

import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=20, help="log2 of the table size")
    ap.add_argument("--batch", type=int, default=1024, help="independent sumchecks per rank per step (32 GiB of tables)")
    ap.add_argument("--transcript", choices=["host", "device"], default="host",
                    help="where MiMC7 runs (host cores between launches, or one GPU lane per sumcheck)")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event timing")
    ap.add_argument("--proofs", type=int, default=64,
                    help="second half of the metric: full GKR proofs of a t.circom-class layered circuit per rank (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (single-GPU boxes): run several ranks on one device over gloo to exercise the N > 1 code path
    backend = os.environ.get("GKR_BENCH_BACKEND", "nccl")
    if "GKR_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["GKR_BENCH_DEVICE"])
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible); there is no CPU fallback")

    from gkr_amd import Context

    n, batch = args.n, args.batch
    count = 1 << n
    ctx = Context(local_rank)
    ctx.set_transcript(1 if args.transcript == "host" else 0)
    tables = ctx.alloc(batch * count * 32)
    for b in range(batch):
        ctx.fill_table(ctypes.c_void_p(tables.value + b * count * 32), count, 0xC0FFEE + 2 + 1000 * rank + b)
    ctx.synchronize()

    step_times = []

    outputs = [None]   # the proof arrays of the previous step are reused (no fresh pages inside the timed call)

    def step():
        t = time.perf_counter()
        outputs[0] = ctx.sumcheck_mle_batch_device(tables, n, batch, out=outputs[0])
        step_times.append(time.perf_counter() - t)
        return outputs[0]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    ctx.profile(0 if args.no_profile else 2)   # on during warm-up too: the event pool is created lazily
    for _ in range(args.warmup):
        step()
    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.profile(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ops_per_sumcheck = 5 * (count - 1)
    total_ops = ops_per_sumcheck * batch * args.steps * world
    value = total_ops / elapsed

    if rank == 0:
        names = ["mle_multifold", "mle_sub_sums", "mle_sub_reduce", "mle_pass_small", "mle_fold_plan",            # multi-round passes (default)
                 "mle_fold_sum", "mle_sum_first", "mle_round_reduce", "mle_fold_sum_small", "mle_round_hash"]   # per-round paths
        prof = {k: ctx.profile_get(k) for k in names}
        dom_name = "mle_multifold" if prof["mle_multifold"]["launches"] else "mle_fold_sum"
        dom = prof[dom_name]
        achieved = dom["bytes"] / (dom["total_ms"] * 1e-3) / 1e9 if dom["total_ms"] > 0 else 0.0
        peak = 8000.0
        kernel_ms_total = sum(v["total_ms"] for v in prof.values())
        sumchecks = batch * args.steps
        line = {
            "metric": "BN254-Fr sumcheck field-ops/sec @ 2^20 vars",
            "value": value,
            "unit": "field-ops/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32x8 (BN254 Fr, 254-bit modular integers; the fold pass's constant-by-table products as exact i8 MFMA with i32 sums)",
            "data": "synthetic",
            "config": {"workload": "plain MLE sumcheck (prove_sumcheck), 2^%d points per table, BASELINE configs[2]" % n,
                       "log2_points": n, "batch_per_gpu": batch, "sumchecks_per_step": batch * world,
                       "transcript": "MiMC7-91 on %s, included in the timed region" % args.transcript,
                       "parallelism": "independent sumchecks per rank, no collective"},
            "sumchecks_per_sec": batch * args.steps * world / elapsed,
            "step_ms_each": [round(x * 1e3, 3) for x in step_times[args.warmup:]],
            "roofline": {
                "bound": "hbm", "kernel": "k_mle_multifold_mfma" if dom_name == "mle_multifold" else "k_" + dom_name,
                "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
                "traffic": None,
                "launches": dom["launches"],
                "avg_launch_us": (dom["total_ms"] * 1e3 / dom["launches"]) if dom["launches"] else None,
                "algorithmic_bytes_per_launch": (dom["bytes"] / dom["launches"]) if dom["launches"] else None,
                "bytes_rule": "fold pass binding J variables: (2^J + 1) * 32 B per output entry (reads 2^J source entries, "
                              "writes one), J = 5 on the large tables; k_mle_fold_sum: 192 B per output pair",
            },
            "kernel_ms": {k: round(v["total_ms"], 3) for k, v in prof.items() if v["launches"]},
            # SURVEY 8d's per-round accounting (128 * 2^n bytes per sumcheck) over ALL kernel time of the step:
            # the multi-round schedule moves fewer bytes than that figure, so this can exceed what HBM delivers
            "survey_accounting_GBps": (128.0 * count * sumchecks) / (kernel_ms_total * 1e-3) / 1e9 if kernel_ms_total else None,
        }
        first = prof["mle_sub_sums"] if prof["mle_sub_sums"]["launches"] else prof["mle_sum_first"]
        if first["total_ms"] > 0:
            line["roofline"]["first_pass_GBps"] = first["bytes"] / (first["total_ms"] * 1e-3) / 1e9
        traffic_file = os.path.join(REPO, "profiles", "r01", "d_pmc_traffic.json")
        if os.path.exists(traffic_file) and dom_name == "mle_multifold":
            try:
                tj = json.load(open(traffic_file))
                if tj.get("batch") == batch and tj.get("n") == n:
                    line["roofline"]["traffic"] = tj["k_mle_multifold_mfma"]["per_launch_mean_bytes"]
                    line["roofline"]["traffic_source"] = "profiles/r01/d_pmc_traffic.json (rocprofv3 --pmc passes of this command)"
            except Exception:
                pass
        if args.proofs > 0:
            # second half of the metric, on its own context, once the sumcheck workload's 32 GiB are released
            ctx.free(tables)
            tables = None
            line["aggregated_proofs"] = proofs_per_sec(local_rank, args.proofs)
            line["aggregated_proofs"]["proofs_per_sec_all_ranks"] = line["aggregated_proofs"]["proofs_per_sec"] * world
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(n, args.cpu_seconds)
        print(json.dumps(line), flush=True)

    if tables is not None:
        ctx.free(tables)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def proofs_per_sec(device, n_proofs):
    """"aggregated proofs/sec": full GKR proofs (gkr_prove = prover::prove, rust/src/gkr/prover.rs:6-96) of
    one layered circuit of the size class SURVEY appendix B.4 estimates for t.circom (4 gate layers,
    k = [5, 6, 7, 7 | input 7]) for `n_proofs` different witnesses (BASELINE configs[3]: 64 inputs).
    Proofs are independent (aggregator.rs:350-355 proves them from a rayon par_iter): one context per
    host thread.  Rank 0 measures its own share; other ranks would do the same work."""
    from gkr_amd import Context, synth
    ks = synth.PROOF_BATCH_KS
    circuit = synth.proof_batch_circuit()
    inputs = synth.proof_batch_witnesses(n_proofs)   # what tests/test_gpu_config_scale.py checks against the oracle
    ctx = Context(device)
    ctx.prove_batch_raw(circuit, inputs[: min(8, n_proofs)])   # warm-up: code objects, workspaces
    ctx.prove_batch_raw(circuit, inputs)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.prove_batch_raw(circuit, inputs)
    dt = (time.perf_counter() - t0) / reps
    ctx.close()
    return {"proofs_per_sec": n_proofs / dt, "proofs": n_proofs, "ms_per_batch": dt * 1e3,
            "circuit": "synthetic layered circuit, 4 gate layers, k = %s, random add/mult gates, %d witnesses" % (ks, n_proofs),
            "how": "gkr_prove_batch: all proofs advance together, every layer sumcheck batched (one round trip per round for all)"}


def usable_cpus():
    """Affinity mask capped by the cgroup CPU quota (the GPU box runs this in a
    container with cpu.max = 16 CPUs although 256 are visible)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p))))
    except Exception:
        pass
    return n


def cpu_baseline(n, seconds):
    """The oracle's dense C prover (a port of the reference algorithm's dense form;
    the reference itself is Rust and cannot be built here) on the host cores."""
    from oracle import cdense
    cores = min(cdense.max_threads(), usable_cpus())
    table = cdense.fill_table(1 << n, 0xC0FFEE + 2)
    cdense.sumcheck_mle_raw(table, n, cores)   # warm-up (constants, page faults)
    done, t0 = 0, time.perf_counter()
    while True:
        cdense.sumcheck_mle_raw(table, n, cores)
        done += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or done >= 2000:
            break
    return {"value": done * 5 * ((1 << n) - 1) / dt, "unit": "field-ops/s", "cores": cores, "kind": "port",
            "sample": "%d sequential 2^%d-point sumchecks (MiMC7 included) in %.1f s, OpenMP over %d threads"
                      % (done, n, dt, cores)}


if __name__ == "__main__":
    main()
