"""gkr_prove_many's LOCKSTEP GROUPS: items whose circuits share their k list are proven together -- one launch per pass for
the group, the passes over the gates reading each proof's own gate lists through a per-proof table -- as the reference's
par_iter proves the (circuit, input) pairs of a step side by side (rust/src/aggregator.rs:411-416).  Same bytes as every item
proven on its own (gkr_prove_batch), which the other suites hold against the CPU checker; spot-checked against the checker
here as well."""
import ctypes

import numpy as np
import pytest

from gkr_amd import Context, GKRCircuit, GkrError, Layer, synth
from gkr_amd import _native as N
from oracle import cdense

pytestmark = pytest.mark.gpu


def _circuit(ks, seed, circom_like=False):
    rng = np.random.default_rng(seed)
    layers = []
    for i in range(len(ks) - 1):
        g, m = 1 << ks[i], 1 << ks[i + 1]
        gt = rng.integers(0, 2, g, dtype=np.uint8)
        l = rng.integers(0, m, g, dtype=np.uint32)
        r = rng.integers(0, m, g, dtype=np.uint32)
        if circom_like:            # relay gates reading one constant wire: ONE heavy right-operand bucket (convert.rs:307-342)
            relay = rng.random(g) < 0.5
            gt[relay] = 0
            r[relay] = 2
        layers.append((gt, l, r))
    return GKRCircuit([Layer(ks[i], *layers[i]) for i in range(len(ks) - 1)], ks[-1]), layers


def _one_by_one(ctx, work):
    """every item through gkr_prove_batch on its own -> per item the nine output arrays"""
    return [[a.copy() for a in ctx.prove_batch_raw(c, x, all_arrays=True)] for c, x in work]


def _assert_same(got, want, tag):
    for j, (g, w) in enumerate(zip(got, want)):
        for n, (a, b) in enumerate(zip(g, w)):
            assert np.array_equal(a, b), (tag, "item", j, "array", n)


@pytest.mark.parametrize("ks_list,batches", [
    ([[3, 4, 5, 4]] * 3 + [[2, 6, 6]] * 2 + [[4, 5]], [2, 1, 3, 1, 1, 2]),          # circom-sized layers: bucket passes
    ([[9, 13, 14]] * 3 + [[10, 15]] * 2, [1, 2, 1, 1, 1]),                          # wide layers: lane-group passes
])
def test_mixed_shape_items_equal_items_proven_alone(ks_list, batches):
    rng = np.random.default_rng(1234)
    work, raw = [], []
    for j, (ks, b) in enumerate(zip(ks_list, batches)):
        c, layers = _circuit(ks, 900 + j, circom_like=(j % 2 == 1))
        work.append((c, np.ascontiguousarray(synth.rand_fr(rng, b << ks[-1]).reshape(b, 1 << ks[-1], 4))))
        raw.append(layers)
    with Context(0) as ctx:
        want = _one_by_one(ctx, work)
        # the CPU checker on one item of every shape
        for j in (0, len(work) - 1):
            ref = cdense.prove_raw(raw[j], work[j][1][0])
            assert np.array_equal(want[j][2][0], np.concatenate(ref["R"]))
        for threads in (0, 1, 3, 6):
            got = ctx.prove_many_raw(ctx.prepare_many(work), threads)
            _assert_same(got, want, ("lockstep", threads))
        ctx.set_option("prove_many_lockstep", 0)
        _assert_same(ctx.prove_many_raw(ctx.prepare_many(work), 0), want, "one chain per item")
        ctx.set_option("prove_many_lockstep", 1)
        ctx.set_option("lockstep_max_proofs", 3)         # groups cut by their proof count
        _assert_same(ctx.prove_many_raw(ctx.prepare_many(work), 0), want, "groups of at most three proofs")


def test_the_same_circuit_twice_in_one_call():
    c, _ = _circuit([3, 5, 5], 77)
    rng = np.random.default_rng(7)
    xs = [np.ascontiguousarray(synth.rand_fr(rng, 2 << 5).reshape(2, 32, 4)) for _ in range(3)]
    work = [(c, x) for x in xs]
    with Context(0) as ctx:
        want = _one_by_one(ctx, work)
        _assert_same(ctx.prove_many_raw(ctx.prepare_many(work), 0), want, "same circuit, three items")


def test_a_bad_item_in_a_group_fails_alone():
    """One witness that does not satisfy its circuit (require_zero_output) and one circuit with a gate out of range, each in
    a group with good items: the bad items get their status, the group's other items are proven all the same."""
    ks = [3, 4, 4]
    rng = np.random.default_rng(99)
    work = []
    for j in range(4):
        c, _ = _circuit(ks, 300 + j)
        work.append((c, np.ascontiguousarray(synth.rand_fr(rng, 1 << ks[-1]).reshape(1, 1 << ks[-1], 4))))
    with Context(0) as ctx:
        want = _one_by_one(ctx, work)
        prepared = ctx.prepare_many(work)
        prepared["items"][1].require_zero_output = 1       # a random witness: output 0 is not zero
        with pytest.raises(GkrError):
            ctx.prove_many_raw(prepared, 0)
        st = [it.status for it in prepared["items"]]
        assert st[1] != 0 and st[0] == st[2] == st[3] == 0, st
        for j in (0, 2, 3):
            assert np.array_equal(prepared["outs"][j][2], want[j][2]), j
        # a gate whose operand is out of range, in a circuit the context has not seen
        bad_c, _ = _circuit(ks, 555)
        bad_c.layer[0].left[0] = 1 << ks[1]
        work2 = work[:3] + [(bad_c, work[3][1])]
        prepared = ctx.prepare_many(work2)
        with pytest.raises(GkrError):
            ctx.prove_many_raw(prepared, 0)
        st = [it.status for it in prepared["items"]]
        assert st[3] == N.GKR_ERR_INVALID and st[:3] == [0, 0, 0], st
        for j in range(3):
            assert np.array_equal(prepared["outs"][j][2], want[j][2]), j


def test_sixteen_sub_circuit_step_in_groups_matches_the_committed_digests():
    """The compiled R1CS the bench's large_r1cs leg proves, at 1/16 of its size (16 384 constraints -> sub-circuits in a
    few shapes): gkr_prove_many in lockstep groups against every item proven alone."""
    from gkr_amd.aggregate import ProvingStep
    from gkr_amd.field import as_limbs
    step = ProvingStep(synth.mimc7_demo_r1cs(nrounds=4096))
    inputs = step.inputs_for(np.stack([as_limbs(synth.mimc7_demo_witness(2, 3, nrounds=4096))]))
    work = list(zip(step.circuits, inputs))
    shapes = {}
    for c in step.circuits:
        shapes.setdefault(tuple(c.get_k_list()), 0)
        shapes[tuple(c.get_k_list())] += 1
    assert max(shapes.values()) >= 2, shapes          # there IS something to group
    with Context(0) as ctx:
        want = _one_by_one(ctx, work)
        for threads in (0, 4):
            _assert_same(ctx.prove_many_raw(ctx.prepare_many(work), threads), want, ("r1cs", threads))
    step.close()


def test_option_changes_between_calls_reach_the_child_contexts_caches():
    """gkr_prove_many keeps child contexts (one per thread) with their own circuit caches between calls.  An option that
    changes the layout of cached gate lists -- gate_groups_min_k decides which layers get the wide-layer item plan,
    gate_segment_log2 / gate_segments_min_log2 the segment form's shift -- set on the PARENT between two calls must drop the
    children's caches too (only the caller's was dropped; the children then proved with lists of the old layout)."""
    rng = np.random.default_rng(4321)
    work = []
    for j, ks in enumerate([[9, 13, 14], [9, 13, 14], [10, 15], [8, 12, 12], [8, 12, 12]]):
        c, _ = _circuit(ks, 700 + j, circom_like=(j % 2 == 1))
        work.append((c, np.ascontiguousarray(synth.rand_fr(rng, 1 << ks[-1]).reshape(1, 1 << ks[-1], 4))))
    with Context(0) as ctx:
        want = _one_by_one(ctx, work)
        ctx.set_option("prove_many_lockstep", 0)           # one chain per item: every child context caches circuits of its own
        _assert_same(ctx.prove_many_raw(ctx.prepare_many(work), 4), want, "defaults")
        for name, value in (("gate_groups_min_k", 12), ("gate_groups_min_k", 16), ("gate_segments_min_log2", 8),
                            ("gate_segment_log2", 2), ("gate_segments_off", 1)):
            ctx.set_option(name, value)
            _assert_same(ctx.prove_many_raw(ctx.prepare_many(work), 4), want, (name, value))
