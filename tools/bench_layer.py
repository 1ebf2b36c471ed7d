#!/usr/bin/env python3
"""GKR-layer sumcheck (prove_sumcheck_opt) timing at BASELINE config-5 scale on one GPU.

    python tools/bench_layer.py --k-i 24 --k 12 --steps 3

Random gates (type in {add, mult}, operands in [0, 2^k)), random z and W (the inputs are
gkr_amd.synth.config5_layer, what tests/test_gpu_config_scale.py checks against the oracle).
Reports per-kernel HIP-event times and the canonical figures: 25 (2^{2k} - 1) field-ops,
256 * 2^{2k} bytes for the sumcheck; the predicate build is reported separately.
No time is printed for a wrong transcript: the sumcheck relations are always checked (g_j(0) + g_j(1) =
g_{j-1}(r_{j-1}), r_j = MiMC7(g_j)), and where tests/golden/config_hashes.json holds the digest of the reference
semantics' transcript for this size, the GPU's transcript must have it."""

import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k-i", type=int, default=24)
    ap.add_argument("--k", type=int, default=12)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--transcript", choices=["host", "device"], default="host")
    ap.add_argument("--resident", action="store_true", help="gate arrays uploaded and sorted once, before the timed calls (gkr_resident_layer_*)")
    args = ap.parse_args()
    from gkr_amd import Context, multi_hash, synth
    from gkr_amd.field import MODULUS as P, from_limbs

    lay, z, W = synth.config5_layer(args.k_i, args.k)
    ctx = Context(0)
    ctx.set_transcript(1 if args.transcript == "host" else 0)
    ctx.profile(True)
    times = []
    gates = None
    if args.resident:
        from gkr_amd.parallel import ResidentGates
        gates = ResidentGates(ctx, args.k_i, 0, *lay.arrays())
    for it in range(args.warmup + args.steps):
        if it == args.warmup:
            ctx.profile_reset()
        t0 = time.perf_counter()
        C, L, R = gates.sumcheck_raw(args.k, z, W) if gates else ctx.sumcheck_layer_raw(lay, args.k, z, W)
        times.append(time.perf_counter() - t0)
    names = ["predicate_scatter", "predicate_normalise", "predicate_sorted",
             "layer_round_fused", "layer_round", "layer_fold", "layer_round_reduce", "layer_round_hash",
             "layer_uv", "layer_uv_round", "layer_collapse", "layer_c_round", "gate_lists", "gate_uv", "gate_rows"]
    prof = {n: ctx.profile_get(n) for n in names}
    steps = args.steps
    N = 1 << (2 * args.k)
    sum_ms = sum(prof[n]["total_ms"] for n in ("layer_round_fused", "layer_round", "layer_fold", "layer_uv", "layer_uv_round",
                                               "layer_collapse", "layer_c_round", "gate_uv", "gate_rows")) / steps
    out = {
        "workload": "GKR layer sumcheck k_i=%d k=%d (2^%d-point hypercube, 2^%d gates)%s" % (
            args.k_i, args.k, 2 * args.k, args.k_i, ", gates resident in HBM" if gates else ", gate arrays uploaded in every call"),
        "wall_ms_per_sumcheck": 1e3 * sum(times[args.warmup:]) / steps,
        "kernel_ms_per_sumcheck": {n: prof[n]["total_ms"] / steps for n in names},
        "field_ops": 25 * (N - 1), "algorithmic_bytes": 256 * N,
        "fused_kernel": {"launches_per_sumcheck": prof["layer_round_fused"]["launches"] / steps,
                         "GBps": prof["layer_round_fused"]["bytes"] / (prof["layer_round_fused"]["total_ms"] * 1e-3) / 1e9
                         if prof["layer_round_fused"]["total_ms"] else None},
        "sumcheck_kernels_GBps": 256 * N / (sum_ms * 1e-3) / 1e9 if sum_ms else None,
        "field_ops_per_s_kernels": 25 * (N - 1) / (sum_ms * 1e-3) if sum_ms else None,
        "field_ops_per_s_wall": 25 * (N - 1) / (sum(times[args.warmup:]) / steps),
    }
    # verifier relations: g_j(0) + g_j(1) = g_{j-1}(r_{j-1}); r_j = MiMC(g_j)
    claim = None
    ok = True
    for j in range(2 * args.k):
        vec = from_limbs(C[j])[3 - int(L[j]):]
        r = from_limbs(R[j])[0]
        if claim is not None and (vec[-1] + sum(vec)) % P != claim:
            ok = False
        if multi_hash(vec) != r:
            ok = False
        acc = 0
        for c in vec:
            acc = (acc * r + c) % P
        claim = acc
    out["verifier_relations_ok"] = ok
    want = synth.golden_digest("layer", "k_i=%d,k=%d" % (args.k_i, args.k))
    out["transcript_sha256"] = synth.transcript_digest(C, L, R)
    out["matches_golden_digest"] = None if want is None else (out["transcript_sha256"] == want)
    ctx.close()
    if os.environ.get("GKR_EXPERIMENT_WRONG_RESULTS_OK") == "1":    # (a variant library built to time a kernel with a piece removed)
        out["EXPERIMENT"] = "results not checked: timing of a deliberately broken variant"
    elif not out["verifier_relations_ok"] or out["matches_golden_digest"] is False:
        raise SystemExit("WRONG TRANSCRIPT (relations ok: %s, golden digest match: %s) -- no timing reported"
                         % (out["verifier_relations_ok"], out["matches_golden_digest"]))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
