"""GPU parity at the sizes BASELINE.json's configs name (bit-exact against the C oracle on the same seeded inputs):

  configs[4]  one GKR layer, k_i = 24, k = 12 (2^24 gates, 2^24-point hypercube): every form of the layer
              sumcheck on one GPU, and the gate-sharded multi-GPU form with 8 logical ranks;
  configs[3]  the exact 64-witness batch of proofs bench.py times, every proof against the oracle and through
              the verifier.

The inputs come from gkr_amd.synth, the module bench.py and tools/bench_layer.py take theirs from.
"""

import os
import subprocess
import sys

import numpy as np
import pytest

from gkr_amd import Context, parallel, synth, verify
from gkr_amd.field import from_limbs
from oracle import cdense

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
K_I, K = 24, 12


@pytest.fixture(scope="module")
def config5_expected(tmp_path_factory):
    """The oracle's transcript of the configs[4] layer (reference semantics: sumcheck.rs:36-156), ~10-20 s of CPU."""
    lay, z, W = synth.config5_layer(K_I, K)
    C, L, R = cdense.sumcheck_layer_raw(K_I, K, lay.gate_type, lay.left, lay.right, z, W)
    path = str(tmp_path_factory.mktemp("c5") / "expected.npz")
    np.savez(path, C=C, L=L, R=R)
    return path, (C, L, R)


@pytest.mark.parametrize("env", [{}, {"GKR_GATE_SEGMENTS_OFF": "1"}, {"GKR_GATE_SEGMENT_LOG2": "6"}, {"GKR_GATE_SORT_GLOBAL": "1"},
                                 {"GKR_GATE_SEGMENTS_NO_LDS": "1"}],
                         ids=["default(gate lists, block-private sort, segment passes)", "bucket passes (no segments)", "segments of 64 gates",
                              "gate lists, global-atomic sort", "segment passes gathering from L2"])
def test_config5_layer_every_form_matches_oracle(config5_expected, env):
    path, _ = config5_expected
    out = subprocess.run([sys.executable, os.path.join(HERE, "config_scale_worker.py"), str(K_I), str(K), path],
                         env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("k_i,k,env", [(16, 8, {"GKR_GATE_SEGMENTS_MIN_LOG2": "16"}), (18, 9, {"GKR_GATE_SEGMENTS_MIN_LOG2": "16"}),
                                       (19, 12, {"GKR_GATE_SEGMENTS_MIN_LOG2": "16"}), (20, 10, {"GKR_GATE_SEGMENTS_MIN_LOG2": "16"}),
                                       (21, 9, {"GKR_GATE_SEGMENTS_MIN_LOG2": "16", "GKR_GATE_SEGMENT_LOG2": "3"}), (22, 11, {}),
                                       (26, 12, {}), (25, 13, {})])
def test_segment_passes_at_other_widths_match_oracle(tmp_path, k_i, k, env):
    """The segment form of the gate passes (csrc/gate_seg.h) where the split of eq(z, .) and the number of sort blocks
    per segment take other values than at configs[4]'s size (one block per segment, many, segments of 8 gates)."""
    lay, z, W = synth.config5_layer(k_i, k)
    C, L, R = cdense.sumcheck_layer_raw(k_i, k, lay.gate_type, lay.left, lay.right, z, W)
    path = str(tmp_path / "expected.npz")
    np.savez(path, C=C, L=L, R=R)
    out = subprocess.run([sys.executable, os.path.join(HERE, "config_scale_worker.py"), str(k_i), str(k), path],
                         env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("scenario", ["skewed", "batch"])
def test_segment_passes_scenarios_match_oracle(scenario):
    """tests/segment_scenarios_worker.py (a child: the size from which layers take the segment passes is read once per
    process, and these layers are below the default):
      skewed  segments far from the average -- every gate of the first quarter of a 2^20-gate layer wired to ONE left
              operand (segments of 2^16 gates, cut into items of 32), a bucket nobody uses, all gates of one type in a stretch;
      batch   gkr_prove_batch over a circuit whose first layer has 2^19 gates: three witnesses advanced together -- every
              proof's own eq tables, partial sums and outputs (the batch index of k_seg_pass / k_seg_combine)."""
    out = subprocess.run([sys.executable, os.path.join(HERE, "segment_scenarios_worker.py"), scenario],
                         env=dict(os.environ, GKR_GATE_SEGMENTS_MIN_LOG2="16"), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


def test_config5_layer_eight_logical_ranks_match_oracle(config5_expected):
    """configs[4] as BASELINE states it: the layer split over 8 ranks.  Eight logical ranks (threads, one context
    each) on the one visible GPU run the gate-sharded form; every rank must hold the oracle's transcript."""
    _, (C, L, R) = config5_expected
    lay, z, W = synth.config5_layer(K_I, K)
    want = ([from_limbs(C[j])[3 - int(L[j]):] for j in range(2 * K)], from_limbs(R))
    got = parallel.prove_sumcheck_opt_logical_gates(0, lay, K, from_limbs(z), from_limbs(W), 8)
    assert all(g == want for g in got)


def test_config4_proof_batch_of_the_bench_matches_oracle():
    """The 64 proofs bench.py's `aggregated_proofs` leg times (same circuit, same witnesses): each one equal to the
    oracle's proof and accepted by the verifier."""
    circuit = synth.proof_batch_circuit()
    inputs = synth.proof_batch_witnesses(64)
    layers = [(l.gate_type, l.left, l.right) for l in circuit.layer]
    with Context(0) as ctx:
        proofs = ctx.prove_batch(circuit, [from_limbs(w) for w in inputs])
    assert len(proofs) == 64
    for b, pr in enumerate(proofs):
        ref = cdense.prove(layers, from_limbs(inputs[b]))
        assert pr.sumcheck_proofs == ref["sumcheck_proofs"] and pr.sumcheck_r == ref["sumcheck_r"], b
        assert pr.q == ref["q"] and pr.z == ref["z"] and pr.r == ref["r"] and pr.k == ref["k"], b
        assert verify(pr, circuit), b


def test_config5_layer_on_resident_gates_matches_oracle(config5_expected):
    """gkr_sumcheck_layer_device: the gate arrays uploaded once (what bench.py --mode layer-split times), whole layer
    and, through the hook with an in-process sum, as two halves."""
    _, (C, L, R) = config5_expected
    lay, z, W = synth.config5_layer(K_I, K)
    gt, l, r = lay.arrays()
    with Context(0) as ctx:
        gates = parallel.ResidentGates(ctx, K_I, 0, gt, l, r)
        for _ in range(2):
            got = gates.sumcheck_raw(K, z, W)
            assert np.array_equal(got[0], C) and np.array_equal(got[1], L) and np.array_equal(got[2], R)
        gates.close()
