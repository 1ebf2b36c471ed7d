"""Child process of tests/test_gpu_config_scale.py: the layer path's schedule knobs are read once per process.
Runs the configs[4] layer sumcheck (gkr_amd.synth.config5_layer) and compares every byte of the transcript with the
expected arrays the parent computed with the C oracle (npz file)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_amd import Context, synth  # noqa: E402


def main():
    k_i, k, path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    want = np.load(path)
    lay, z, W = synth.config5_layer(k_i, k)
    with Context(0) as ctx:
        C, L, R = ctx.sumcheck_layer_raw(lay, k, z, W)
    if not (np.array_equal(C, want["C"]) and np.array_equal(L, want["L"]) and np.array_equal(R, want["R"])):
        bad = [j for j in range(2 * k) if not (np.array_equal(C[j], want["C"][j]) and np.array_equal(R[j], want["R"][j]))]
        print("MISMATCH in rounds", bad)
        return 1
    print("OK")
    return 0


if __name__ == "__main__":
    sys.exit(main())
