#!/usr/bin/env python3
"""Golden digests of the oracle's transcripts at BASELINE config sizes (too large to commit as vectors).

    python tests/golden/make_config_hashes.py      # rewrites tests/golden/config_hashes.json (~1 min of CPU)
    python tests/golden/make_config_hashes.py --bench-batch   # rewrites tests/golden/bench_batch_hashes.json (~6 min, 8 cores)

For each workload of gkr_amd.synth the C oracle (oracle/c, pinned against the reference-generated fixtures by
tests/test_oracle_*.py) produces the transcript; what is committed is sha256 over the raw output arrays
(coefficients | lengths | challenges, the C ABI's layout).  bench.py and tools/bench_layer.py compare the GPU
transcript of the workload they time against these digests without running the oracle.
"""
import hashlib
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from gkr_amd import synth  # noqa: E402
from oracle import cdense  # noqa: E402


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def _bench_table_transcript(seed):
    """One table of bench.py's default workload: fill_table(2^20, seed) -> the oracle's (C, L, R) as bytes, and the
    transcripts of its sixteen 2^16-point slices (what the n16 leg proves on the same memory)."""
    table = cdense.fill_table(1 << 20, seed)
    C, L, R = cdense.sumcheck_mle_raw(table, 20, 1)
    slices = [cdense.sumcheck_mle_raw(table[j << 16:(j + 1) << 16], 16, 1) for j in range(16)]
    return seed, (C.tobytes(), L.tobytes(), R.tobytes()), [(c.tobytes(), l.tobytes(), r.tobytes()) for c, l, r in slices]


def bench_batch_digests(ranks=8, batch=1024, n16_tables=4096):
    """Digests of everything bench.py's default line times (gkr_amd.synth.bench_table_seed): per rank the whole batch of
    1024 x 2^20 transcripts (sha256 over the C | L | R output arrays as the C ABI lays them out for the batch), rank 0's
    tables one by one (first 16 hex digits), and rank 0's 4096 x 2^16 leg."""
    import multiprocessing
    seeds = sorted({synth.bench_table_seed(r, b) for r in range(ranks) for b in range(batch)})
    with multiprocessing.Pool(min(8, os.cpu_count() or 1)) as pool:
        done = {}
        n16 = {}
        for seed, t20, t16 in pool.imap_unordered(_bench_table_transcript, seeds, chunksize=4):
            done[seed] = t20
            if seed < synth.bench_table_seed(0, 0) + n16_tables // 16:
                n16[seed] = t16
    out = {"batch": batch, "n": 20, "whole_batch_by_rank": {}, "rank0_tables": [], "n16_whole_batch_rank0": None, "n16_tables": n16_tables}
    for r in range(ranks):
        hc, hl, hr = [], [], []
        for b in range(batch):
            c, l, rr = done[synth.bench_table_seed(r, b)]
            hc.append(c), hl.append(l), hr.append(rr)
        out["whole_batch_by_rank"][str(r)] = hashlib.sha256(b"".join(hc) + b"".join(hl) + b"".join(hr)).hexdigest()
    for b in range(batch):
        out["rank0_tables"].append(hashlib.sha256(b"".join(done[synth.bench_table_seed(0, b)])).hexdigest()[:16])
    hc, hl, hr = [], [], []
    for j in range(n16_tables):
        c, l, rr = n16[synth.bench_table_seed(0, j >> 4)][j & 15]
        hc.append(c), hl.append(l), hr.append(rr)
    out["n16_whole_batch_rank0"] = hashlib.sha256(b"".join(hc) + b"".join(hl) + b"".join(hr)).hexdigest()
    return out


def _demo_digests(pair):
    from oracle import convert as oconv
    a, b = pair
    r = _demo_digests.r1cs
    return [synth.proof_digest(p["sumcheck_proofs"], p["sumcheck_r"], p["q"], p["z"], p["r"])[:16]
            for p in (cdense.prove(sub["layers"], sub["input_values"]) for sub in oconv.convert_r1cs_wtns_gkr(r, synth.mimc7_demo_witness(a, b)))]


def proof_digests(n_inputs=64):
    """The CPU checker's proofs (oracle/convert.py + the dense C prover, both pinned by the reference's fixtures / the
    hand-derived circuits) of the demo circuit's 12 sub-circuits: configs[0]'s three example inputs and the 64 inputs of
    configs[3], one 16-hex-digit digest per (input, sub-circuit)."""
    from oracle import convert as oconv
    _demo_digests.r1cs = oconv.read_r1cs(synth.mimc7_demo_r1cs().serialize())
    # (serially: ~10 s; a process pool forked after the C checker's OpenMP runtime has started hangs)
    ex = [_demo_digests(p) for p in synth.EXAMPLE_INPUTS]
    many = [_demo_digests(p) for p in synth.demo_proof_inputs(n_inputs)]
    return {"what": "sha256[:16] of gkr_amd.synth.proof_digest over (sumcheck_proofs, sumcheck_r, q, z, r) per (input, sub-circuit)",
            "config0": {"inputs": [list(p) for p in synth.EXAMPLE_INPUTS], "digests": ex},
            "config3": {"inputs": n_inputs, "digests": many}}


def large_r1cs_digests(nrounds=65536, pair=(2, 3)):
    """A 262 144-constraint R1CS (gkr_amd.synth.mimc7_demo_r1cs with 65 536 rounds: the demo's constraint shapes, the
    constants repeating) through the CPU checker END TO END: oracle/convert.py's restatement of convert.rs compiles it into
    its layered circuits (layers of 2^14 .. 2^16 values), the linear-time C prover proves each for one witness; per sub-circuit
    the k list and the digest of the proof's arrays (gkr_amd.synth.proof_arrays_digest)."""
    import time
    from oracle import convert as oconv
    t = time.time()
    r = oconv.read_r1cs(synth.mimc7_demo_r1cs(nrounds=nrounds).serialize())
    subs = oconv.convert_r1cs_wtns_gkr(r, synth.mimc7_demo_witness(pair[0], pair[1], nrounds=nrounds))
    print("oracle compile: %.0f s, %d sub-circuits" % (time.time() - t, len(subs)), flush=True)
    out = {"what": "sha256 of gkr_amd.synth.proof_arrays_digest per sub-circuit; compile and proofs by the CPU checker",
           "nrounds": nrounds, "constraints": 4 * nrounds, "input": list(pair), "k": [], "digests": [], "coeff_digests": []}
    for sub in subs:
        layers = [(np.asarray(a, dtype=np.uint8), np.asarray(b, dtype=np.uint32), np.asarray(c, dtype=np.uint32)) for a, b, c in sub["layers"]]
        ref = cdense.prove_raw(layers, cdense.to_limbs(sub["input_values"]))
        out["k"].append(list(sub["k"]))
        out["digests"].append(synth.proof_arrays_digest(list(sub["k"]), *synth.proof_arrays_from_checker(ref, list(sub["k"]))))
        out["coeff_digests"].append(synth.proof_coeffs_digest(cdense.mobius_raw(ref["values"][0], ref["k"][0]), cdense.mobius_raw(ref["values"][-1], ref["k"][-1])))
    return out


def main():
    if "--large-r1cs" in sys.argv:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "large_r1cs_digests.json")
        with open(path, "w") as f:
            json.dump(large_r1cs_digests(), f, indent=0, sort_keys=True)
            f.write("\n")
        print("wrote", path)
        return
    if "--proofs" in sys.argv:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "proof_digests.json")
        with open(path, "w") as f:
            json.dump(proof_digests(), f, indent=0, sort_keys=True)
            f.write("\n")
        print("wrote", path)
        return
    if "--circom-shaped" in sys.argv:   # only the compiler-shaped wide layer's digest, added to the committed file
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_hashes.json")
        cur = json.load(open(path))
        for k_i, k in ((20, 20),):
            lay, z, W = synth.circom_shaped_layer(k_i, k)
            C, L, R = cdense.sumcheck_layer_lin_raw(k_i, k, lay.gate_type, lay.left, lay.right, z, W)
            cur["layer"]["circom-shaped,k_i=%d,k=%d" % (k_i, k)] = digest(C, L, R)
        with open(path, "w") as f:
            json.dump(cur, f, indent=1, sort_keys=True)
            f.write("\n")
        print("wrote", path)
        return
    if "--prove-coeffs" in sys.argv:   # only the d / input_func digests of the wide proof, added to the committed file
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_hashes.json")
        cur = json.load(open(path))
        cur["prove_coeffs"] = {}
        for ks in ((18, 20, 20),):
            circuit, raw, wit = synth.wide_circuit(ks)
            ref = cdense.prove_raw(raw, wit[0])
            assert cur["prove"]["k=" + ",".join(map(str, ks))] == synth.proof_arrays_digest(list(ks), *synth.proof_arrays_from_checker(ref, list(ks)))
            cur["prove_coeffs"]["k=" + ",".join(map(str, ks))] = synth.proof_coeffs_digest(cdense.mobius_raw(ref["values"][0], ks[0]), cdense.mobius_raw(ref["values"][-1], ks[-1]))
        with open(path, "w") as f:
            json.dump(cur, f, indent=1, sort_keys=True)
            f.write("\n")
        print("wrote", path)
        return
    out = {"what": "sha256(coeffs | lens | challenges) of the C oracle's transcript; inputs: gkr_amd.synth", "layer": {}, "mle": {}}
    if "--bench-batch" in sys.argv:   # ~6 min on 8 cores: kept in its own file, regenerated only when asked
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_batch_hashes.json")
        with open(path, "w") as f:
            json.dump(bench_batch_digests(), f, indent=0, sort_keys=True)
            f.write("\n")
        print("wrote", path)
        return
    for k_i, k in ((16, 8), (20, 10), (24, 12)):
        lay, z, W = synth.config5_layer(k_i, k)
        C, L, R = cdense.sumcheck_layer_raw(k_i, k, lay.gate_type, lay.left, lay.right, z, W)
        out["layer"]["k_i=%d,k=%d" % (k_i, k)] = digest(C, L, R)
    # a WIDE layer (2^20 gates over a next layer of 2^20 values: the shape of a compiled R1CS's big layers), by the linear-time
    # C prover -- the dense form cannot follow
    for k_i, k in ((20, 20),):
        lay, z, W = synth.config5_layer(k_i, k)
        C, L, R = cdense.sumcheck_layer_lin_raw(k_i, k, lay.gate_type, lay.left, lay.right, z, W)
        out["layer"]["k_i=%d,k=%d" % (k_i, k)] = digest(C, L, R)
    # a whole proof through wide layers (gkr_amd.synth.wide_circuit): the limb-only CPU prover on the linear-time layer form
    out["prove"] = {}
    for ks in ((18, 20, 20),):
        circuit, raw, wit = synth.wide_circuit(ks)
        ref = cdense.prove_raw(raw, wit[0])
        out["prove"]["k=" + ",".join(map(str, ks))] = synth.proof_arrays_digest(list(ks), *synth.proof_arrays_from_checker(ref, list(ks)))
        out["prove_coeffs"] = out.get("prove_coeffs", {})
        out["prove_coeffs"]["k=" + ",".join(map(str, ks))] = synth.proof_coeffs_digest(cdense.mobius_raw(ref["values"][0], ks[0]), cdense.mobius_raw(ref["values"][-1], ks[-1]))
    for n, seed in ((16, synth.SEED + 1), (20, synth.SEED + 2)):
        C, L, R = cdense.sumcheck_mle_raw(cdense.fill_table(1 << n, seed), n)
        out["mle"]["n=%d,seed=%d" % (n, seed)] = digest(C, L, R)
    # the tables bench.py --mode mle-split splits over the ranks (seed SEED + 2 at every size); 2^30 points = 32 GiB of
    # table: in place, only with --huge (kept from the committed file otherwise)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_hashes.json")
    try:
        old = json.load(open(path))["mle"]
    except (OSError, KeyError, ValueError):
        old = {}
    for n in (24, 27, 30):
        key = "n=%d,seed=%d" % (n, synth.SEED + 2)
        if n == 30 and "--huge" not in sys.argv:
            if key in old:
                out["mle"][key] = old[key]
            continue
        C, L, R = cdense.sumcheck_mle_inplace_raw(cdense.fill_table(1 << n, synth.SEED + 2), n)
        out["mle"][key] = digest(C, L, R)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_hashes.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
