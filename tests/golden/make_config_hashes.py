#!/usr/bin/env python3
"""Golden digests of the oracle's transcripts at BASELINE config sizes (too large to commit as vectors).

    python tests/golden/make_config_hashes.py      # rewrites tests/golden/config_hashes.json (~1 min of CPU)

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


def main():
    out = {"what": "sha256(coeffs | lens | challenges) of the C oracle's transcript; inputs: gkr_amd.synth", "layer": {}, "mle": {}}
    for k_i, k in ((16, 8), (20, 10), (24, 12)):
        lay, z, W = synth.config5_layer(k_i, k)
        C, L, R = cdense.sumcheck_layer_raw(k_i, k, lay.gate_type, lay.left, lay.right, z, W)
        out["layer"]["k_i=%d,k=%d" % (k_i, k)] = digest(C, L, R)
    for n, seed in ((16, synth.SEED + 1), (20, synth.SEED + 2)):
        C, L, R = cdense.sumcheck_mle_raw(cdense.fill_table(1 << n, seed), n)
        out["mle"]["n=%d,seed=%d" % (n, seed)] = digest(C, L, R)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_hashes.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
