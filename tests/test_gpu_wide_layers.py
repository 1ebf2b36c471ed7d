"""GPU parity for WIDE layers: next-layer tables of 2^14 .. 2^20 values and more, which prover::prove accepts like any
other GKRCircuit (rust/src/gkr/prover.rs:23-83; layers are padded to whatever 2^k the compiler needs,
rust/src/convert.rs:209-214).  The checker is the C oracle's linear-time layer prover (oracle/c/ogkr.c,
ogkr_sumcheck_layer_lin), which tests/test_oracle_c.py pins against the dense forms and the reference's fixtures for
every width those reach.  Bit-exact: every byte of (coefficients, lengths, challenges) and of q, z, r, d, input_func."""

import os
import subprocess
import sys

import numpy as np
import pytest

from gkr_amd import Context, GKRCircuit, GkrError, Layer, synth
from oracle import cdense

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def _same(got, want):
    return all(np.array_equal(a, b) for a, b in zip(got, want))


@pytest.mark.parametrize("k_i,k", [(20, 15), (22, 16), (24, 18), (20, 20), (14, 14), (10, 17), (21, 13), (0, 14), (5, 19), (26, 14), (23, 23)])
def test_wide_layer_sumcheck_matches_oracle(ctx, k_i, k):
    lay, z, W = synth.config5_layer(k_i, k, seed=synth.SEED + 100 * k_i + k)
    want = cdense.sumcheck_layer_lin_raw(k_i, k, lay.gate_type, lay.left, lay.right, z, W)
    got = ctx.sumcheck_layer_raw(lay, k, z, W)
    assert _same(got, want)


@pytest.mark.parametrize("k_i,k,edge", [(12, 15, "random"), (13, 16, "extremes"), (12, 17, "random"), (14, 19, "random"), (16, 20, "random"), (10, 20, "extremes"), (8, 17, "zeros"),
                                        (18, 21, "random")])
def test_matrix_core_product_passes_match_the_checker_and_the_valu_form(k_i, k, edge):
    """Product passes over tables of 2^15 entries and more run on int8 MFMA (mfma_cross.h: the 64 cross sums as byte-digit
    matrix products in blocks of 128 .. 2048 entries per sub-block -- k = 15 .. 21 -- and, from 2^17 entries, the pending fold
    of a phase's later passes through the fold passes' digit matrices).  The same bytes as the checker and as
    the option's other value; W of zeros, and W of r - 1 and 2^253-ish entries (every byte digit at its extremes, the
    anti-diagonal sums at their largest)."""
    lay, z, W = synth.config5_layer(k_i, k, seed=9100 + 10 * k_i + k)
    W = np.ascontiguousarray(W).copy()
    if edge == "zeros":
        W[:] = 0
        W[5] = np.array([1, 0, 0, 0], dtype=np.uint64)
    elif edge == "extremes":
        r_minus_1 = np.array([0x43E1F593F0000000, 0x2833E84879B97091, 0xB85045B68181585D, 0x30644E72E131A029], dtype=np.uint64)
        all_ff = np.array([0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF, 0x2FFFFFFFFFFFFFFF], dtype=np.uint64)
        W[0::2] = r_minus_1
        W[1::4] = all_ff
    want = cdense.sumcheck_layer_lin_raw(k_i, k, lay.gate_type, lay.left, lay.right, z, W)
    with Context(0) as a, Context(0) as b, Context(0) as c:
        b.set_option("no_mfma_cross", 1)
        c.set_option("prod_fold_min_log2", 14 if k < 20 else 20)   # (every later pass folds on the matrix cores / only the second)
        assert a.get_option("no_mfma_cross") == 0
        for ctx_ in (a, b, c):
            assert _same(ctx_.sumcheck_layer_raw(lay, k, z, W), want)


def test_resident_layer_with_W_in_device_memory(ctx):
    """gkr_resident_layer_sumcheck_wdev: W already on the device (what prover::prove has) -- the same transcript as with W
    handed over in host memory, a non-canonical W entry is reported, and the compiler-shaped layer of the bench's
    wide20_circom_shaped leg (at a size the test affords) matches the checker."""
    from gkr_amd import parallel
    for (k_i, k), shaped in (((16, 15), False), ((17, 17), True), ((9, 5), False)):
        lay, z, W = (synth.circom_shaped_layer(k_i, k, seed=77) if shaped else synth.config5_layer(k_i, k, seed=4100 + k))
        gt, l, r = lay.arrays()
        want = cdense.sumcheck_layer_lin_raw(k_i, k, gt, l, r, z, W)
        gates = parallel.ResidentGates(ctx, k_i, 0, gt, l, r)
        assert _same(gates.sumcheck_raw(k, z, W), want)
        d_W = ctx.alloc(W.nbytes)
        ctx.upload(d_W, np.ascontiguousarray(W))
        for _ in range(2):
            assert _same(gates.sumcheck_raw_device_w(k, z, d_W), want)
        bad = np.ascontiguousarray(W).copy()
        bad[3] = np.array([0xFFFFFFFFFFFFFFFF] * 4, dtype=np.uint64)
        ctx.upload(d_W, bad)
        with pytest.raises(GkrError) as e:
            gates.sumcheck_raw_device_w(k, z, d_W)
        assert e.value.status == 2   # GKR_ERR_NON_CANONICAL
        ctx.free(d_W)
        gates.close()


def _circom_like_layer(k_i, k, seed):
    """What a compiled layer looks like (convert.rs:278-343): half of the gates are relay gates Add(x, zero) that all read
    the zero slot as their right operand -- ONE bucket with 2^(k_i - 1) gates --, a few hot wires feed thousands of gates,
    the rest is scattered; a stretch of one gate type; a slot nobody reads."""
    rng = np.random.default_rng(seed)
    g = 1 << k_i
    gt = rng.integers(0, 2, g, dtype=np.uint8)
    l = rng.integers(0, 1 << k, g, dtype=np.uint32)
    r = rng.integers(0, 1 << k, g, dtype=np.uint32)
    relay = rng.random(g) < 0.5
    gt[relay] = 0
    r[relay] = 2                                   # the lazily allocated zero slot
    hot = rng.random(g) < 0.05
    l[hot] = rng.integers(0, 4, int(hot.sum()), dtype=np.uint32) * 1000 + 17   # four wires with ~1 % of the gates each
    l[l == 5] = 6                                  # bucket 5 empty
    gt[: g >> 3] = 1
    return Layer(k_i, gt, l, r)


@pytest.mark.parametrize("k_i,k", [(18, 18), (20, 19), (16, 14)])
def test_wide_layer_with_heavy_buckets_matches_oracle(ctx, k_i, k):
    lay = _circom_like_layer(k_i, k, 31 * k_i + k)
    rng = np.random.default_rng(5)
    z, W = synth.rand_fr(rng, k_i), synth.rand_fr(rng, 1 << k)
    want = cdense.sumcheck_layer_lin_raw(k_i, k, lay.gate_type, lay.left, lay.right, z, W)
    assert _same(ctx.sumcheck_layer_raw(lay, k, z, W), want)


def test_wide_layer_short_round_vectors(ctx):
    """W without some of its variables (get_univariate_coeff, poly.rs:388-420: the round vector then has two entries): the
    dependence flags of a wide table are found over a grid (k_depends_wide)."""
    k_i, k = 16, 15
    lay, z, _ = synth.config5_layer(k_i, k, seed=9)
    rng = np.random.default_rng(10)
    base = synth.rand_fr(rng, 1 << 5)
    idx = np.arange(1 << k)
    for W in (base[(idx >> 3) & 31],                     # depends on variables 8 .. 12 only
              base[idx & 31],                             # the last five
              np.repeat(base[:1], 1 << k, axis=0)):       # constant
        W = np.ascontiguousarray(W)
        want = cdense.sumcheck_layer_lin_raw(k_i, k, lay.gate_type, lay.left, lay.right, z, W)
        assert set(want[1].tolist()) <= {2, 3} and (want[1] == 2).any()
        assert _same(ctx.sumcheck_layer_raw(lay, k, z, W), want)


def _random_circuit(ks, seed):
    rng = np.random.default_rng(seed)
    layers = []
    for i in range(len(ks) - 1):
        g, m = 1 << ks[i], 1 << ks[i + 1]
        layers.append((rng.integers(0, 2, g, dtype=np.uint8), rng.integers(0, m, g, dtype=np.uint32), rng.integers(0, m, g, dtype=np.uint32)))
    return layers


def _check_proofs(arrs, layers, ks, witnesses):
    sc, sl, sr, q, ql, z, rr, dco, ico = arrs
    for b in range(witnesses.shape[0]):
        ref = cdense.prove_raw(layers, witnesses[b])
        ro = qo = 0
        zo = ks[0]
        assert not z[b, :ks[0]].any()                    # z[0] = 0 (prover.rs:16-21)
        for i in range(len(layers)):
            k = ks[i + 1]
            assert np.array_equal(sc[b, ro:ro + 2 * k], ref["C"][i]), (b, i)
            assert np.array_equal(sl[b, ro:ro + 2 * k], ref["L"][i]), (b, i)
            assert np.array_equal(sr[b, ro:ro + 2 * k], ref["R"][i]), (b, i)
            assert int(ql[b, i]) == ref["q_len"][i] and np.array_equal(q[b, qo:qo + k + 1], ref["q"][i]), (b, i)
            assert np.array_equal(z[b, zo:zo + k], ref["z"][i + 1]), (b, i)
            ro += 2 * k
            qo += k + 1
            zo += k
        assert np.array_equal(rr[b], ref["r"])
        assert np.array_equal(dco[b], cdense.mobius_raw(ref["values"][0], ks[0]))
        assert np.array_equal(ico[b], cdense.mobius_raw(ref["values"][-1], ks[-1]))


def test_prove_with_an_input_layer_of_2pow18_values(ctx):
    """A whole proof (prover::prove, prover.rs:6-96) through layers of 2^16 and 2^18 values: sumchecks, q (the stepwise line
    restriction with its grid set-up), z, r, and d / input_func from the device's Moebius transform."""
    ks = [12, 16, 18]
    layers = _random_circuit(ks, 71)
    circuit = GKRCircuit([Layer(ks[i], *layers[i]) for i in range(len(layers))], ks[-1])
    wit = synth.rand_fr(np.random.default_rng(72), 1 << ks[-1])[None]
    _check_proofs(ctx.prove_batch_raw(circuit, wit, all_arrays=True), layers, ks, wit)


def test_prove_batch_of_three_through_wide_layers(ctx):
    ks = [13, 15, 14, 16]
    layers = _random_circuit(ks, 81)
    circuit = GKRCircuit([Layer(ks[i], *layers[i]) for i in range(len(layers))], ks[-1])
    wit = np.stack([synth.rand_fr(np.random.default_rng(820 + b), 1 << ks[-1]) for b in range(3)])
    arrs = ctx.prove_batch_raw(circuit, wit, all_arrays=True)
    _check_proofs(arrs, layers, ks, wit)
    # and again from the circuit cache (sorted lists and heavy-bucket work lists kept per circuit)
    arrs2 = ctx.prove_batch_raw(circuit, wit[:2], all_arrays=True)
    assert all(np.array_equal(a[:2], b) for a, b in zip(arrs, arrs2))


@pytest.mark.parametrize("k_i,k", [(20, 15), (20, 20), (22, 16)])
def test_wide_layer_batch_of_three(ctx, k_i, k):
    """Three witnesses through a circuit whose first layer is (k_i, k): gkr_prove_batch advances the three proofs together
    (every lane-group pass, heavy-bucket unit, product pass and exchange-free hand-off with its batch index)."""
    ks = [k_i, k, 8]
    layers = _random_circuit(ks, 500 + k_i + k)
    circuit = GKRCircuit([Layer(ks[i], *layers[i]) for i in range(len(layers))], ks[-1])
    wit = np.stack([synth.rand_fr(np.random.default_rng(830 + b), 1 << ks[-1]) for b in range(3)])
    _check_proofs(ctx.prove_batch_raw(circuit, wit, all_arrays=True), layers, ks, wit)


@pytest.mark.parametrize("seed", range(6))
def test_random_circuits_across_the_width_thresholds(ctx, seed):
    """Random circuits whose layer widths straddle every switch of the path -- k = 12 / 13 (block-per-bucket vs lane-group gate
    passes, one-block vs grid Moebius and line set-up), 13 / 14 (dependence flags in the prologue vs over a grid, the eq table's
    split form), 2^16 values (host vs device validation and upload), narrow layers between wide ones, a one-gate output layer,
    batches of 1 .. 3 -- every array of every proof against the CPU checker's."""
    rng = np.random.default_rng(9000 + seed)
    depth = int(rng.integers(2, 5))
    ks = [int(rng.integers(0, 15))] + [int(rng.integers(11, 17)) if rng.random() < 0.7 else int(rng.integers(1, 8)) for _ in range(depth)]
    layers = _random_circuit(ks, 9100 + seed)
    if seed % 3 == 0:          # a constant wire every second gate reads, as relay gates do
        gt, l, r = (a.copy() for a in layers[-1])
        r[::2] = 1
        gt[::2] = 0
        layers[-1] = (gt, l, r)
    circuit = GKRCircuit([Layer(ks[i], *layers[i]) for i in range(len(layers))], ks[-1])
    batch = 1 + seed % 3
    wit = np.stack([synth.rand_fr(np.random.default_rng(9200 + 10 * seed + b), 1 << ks[-1]) for b in range(batch)])
    if seed == 4:              # an input layer that does not depend on its last variables: short round vectors, short q
        wit = np.repeat(wit[:, ::4], 4, axis=1)
    _check_proofs(ctx.prove_batch_raw(circuit, np.ascontiguousarray(wit), all_arrays=True), layers, ks, np.ascontiguousarray(wit))


@pytest.mark.parametrize("scenario", ["small", "heavy-batch"])
def test_lane_group_passes_on_small_layers(scenario):
    """tests/wide_scenarios_worker.py (a child: GKR_GATE_GROUPS_MIN_K is read once per process): the lane-group form of the
    gate passes and its heavy-bucket units forced onto layers the dense C oracle reaches, also in a batch of proofs."""
    env = dict(os.environ, GKR_GATE_GROUPS_MIN_K="1")
    out = subprocess.run([sys.executable, os.path.join(HERE, "wide_scenarios_worker.py"), scenario], env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


def test_line_restriction_of_wide_layers_in_whole_proofs():
    """q (and its length) of layers of 2^13 .. 2^15 values -- the line restriction's launches over a grid, an even and an odd
    number of them, then its one-block tail -- in whole proofs against the C checker."""
    out = subprocess.run([sys.executable, os.path.join(HERE, "wide_scenarios_worker.py"), "prove-wide"], env=dict(os.environ),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("nshards,device_exchange", [(3, True), (2, False)])
def test_wide_layer_split_by_gates_over_logical_ranks(nshards, device_exchange):
    """The gate-sharded form (two sum-over-ranks exchanges of 2 * 2^k field elements) on a wide layer: every rank's
    lane-group passes over its own gates, exchange on the device or through the host hook."""
    from gkr_amd import parallel
    k_i, k = 17, 15
    lay = _circom_like_layer(k_i, k, 99)
    rng = np.random.default_rng(6)
    z, W = synth.rand_fr(rng, k_i), synth.rand_fr(rng, 1 << k)
    want = cdense.sumcheck_layer_lin_raw(k_i, k, lay.gate_type, lay.left, lay.right, z, W)
    if device_exchange:
        got = parallel.prove_sumcheck_opt_logical_gates_dev(0, lay, k, z, W, nshards)
        assert len(got) == nshards and all(_same(g, want) for g in got)
    else:
        from gkr_amd.field import from_limbs
        got = parallel.prove_sumcheck_opt_logical_gates(0, lay, k, from_limbs(z), from_limbs(W), nshards)
        proof = [from_limbs(want[0][j])[3 - int(want[1][j]):] for j in range(2 * k)]
        assert len(got) == nshards and all(g == (proof, from_limbs(want[2])) for g in got)


def test_wide_W_is_validated_on_the_device(ctx):
    """A W of 2^16 entries and more is copied by the copy engine and checked for entries >= r where it lands (the reference
    unwrap()s from_repr, sumcheck.rs:16,21): GKR_ERR_NON_CANONICAL, and the context keeps working."""
    from gkr_amd import parallel
    from gkr_amd import _native as N
    k_i, k = 12, 16
    lay, z, W = synth.config5_layer(k_i, k, seed=3)
    gates = parallel.ResidentGates(ctx, k_i, 0, *lay.arrays())
    try:
        bad = W.copy()
        bad[12345] = np.array([0xffffffffffffffff] * 4, dtype=np.uint64)
        with pytest.raises(GkrError) as e:
            gates.sumcheck_raw(k, z, bad)
        assert e.value.status == N.GKR_ERR_NON_CANONICAL
        want = cdense.sumcheck_layer_lin_raw(k_i, k, lay.gate_type, lay.left, lay.right, z, W)
        assert _same(gates.sumcheck_raw(k, z, W), want)
    finally:
        gates.close()


def test_limits_are_reported(ctx):
    lay = Layer(2, [0, 1, 0, 1], [0, 1, 2, 3], [3, 2, 1, 0])
    with pytest.raises(GkrError) as e:
        ctx.sumcheck_layer_raw(lay, 25, np.zeros((2, 4), dtype=np.uint64), np.zeros((1 << 25, 4), dtype=np.uint64))
    assert "GKR_MAX_K_NEXT" in str(e.value)


def test_library_verifier_on_gpu_proofs_through_wide_layers(ctx):
    """gkr_verify (csrc/dropin.cpp; the relations of python/gkr.py:202-231 on the gkr_proof_buf) accepts what gkr_prove_batch
    produced for a circuit with layers of 2^13 .. 2^15 values -- the interpreted verifier cannot follow at that size -- for
    every proof of a batch of three, with one and with many threads; a changed round coefficient, challenge, q entry, r*, z
    entry, input coefficient or gate type is refused with the check that names it."""
    from gkr_amd.dropin import verify_native
    ks = [12, 14, 15, 13]
    circuit, layers, _ = synth.wide_circuit(ks, seed=4242)
    rng = np.random.default_rng(8)
    wit = np.ascontiguousarray(synth.rand_fr(rng, 3 << ks[-1]).reshape(3, 1 << ks[-1], 4))
    arrs = [a.copy() for a in ctx.prove_batch_raw(circuit, wit, all_arrays=True)]
    for b in range(3):
        for threads in (1, 0):
            assert verify_native(circuit, arrs, index=b, threads=threads) == (True, 0, 0), (b, threads)
    rows1 = 2 * ks[1]                      # rounds of layer 0; layer 1's rows follow
    for name, arr, index, check, layer in (("round coefficient", 0, (1, rows1 + 5, 2, 0), 4, 1), ("challenge", 2, (1, 3, 0), 5, 0),
                                           ("q", 3, (1, ks[1] + 1 + 4, 0), 6, 1), ("r*", 6, (1, 2, 0), 7, 2), ("z", 5, (1, ks[0] + 1, 0), 8, 0),
                                           ("input_func", 8, (1, 77, 0), 9, 3)):
        bad = [a.copy() for a in arrs]
        bad[arr][index] ^= np.uint64(1)
        ok, at, why = verify_native(circuit, bad, index=1, threads=0)
        assert not ok and why == check and at == layer, (name, ok, at, why)
        assert verify_native(circuit, bad, index=0, threads=0)[0]          # (the other proofs of the batch are untouched)
    flipped = [(gt.copy(), l, r) for gt, l, r in layers]
    flipped[2][0][123] ^= 1
    from gkr_amd import GKRCircuit, Layer
    wrong = GKRCircuit([Layer(ks[i], *flipped[i]) for i in range(3)], ks[-1])
    assert verify_native(wrong, arrs, index=2, threads=0)[:1] == (False,)
