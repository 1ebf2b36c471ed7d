"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and
the committed golden fixtures.  Bit-exact: this is integer arithmetic.

Run on the GPU box with `python -m pytest tests -m gpu`.
"""

import random

import numpy as np
import pytest

import gkr_amd
from gkr_amd import Context, GKRCircuit, GkrError, Layer
from gkr_amd import _native as N
from oracle import cdense, dense
from oracle.field import P
from helpers import ints, layers_of, right_aligned_equal, terms_as_set

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["host", "device"])
def ctx(request):
    """Every parity test runs with both transcript placements: MiMC7 on the host
    cores between launches (default) and MiMC7 on the device."""
    c = Context(0)
    c.set_transcript(N.GKR_TRANSCRIPT_HOST if request.param == "host" else N.GKR_TRANSCRIPT_DEVICE)
    c._mode = request.param
    yield c
    c.close()


def _circuit(layers, n_inputs):
    ks = [max(0, (len(l[0]) - 1).bit_length()) for l in layers] + [max(0, (n_inputs - 1).bit_length())]
    return GKRCircuit([Layer(ks[i], *layers[i]) for i in range(len(layers))], ks[-1])


# ---------------------------------------------------------------- plain MLE sumcheck

def test_device_reports_gfx950(ctx):
    assert "gfx950" in ctx.device_name()


def test_mle_golden_fixtures(ctx, mle_cases):
    for case in mle_cases:
        proof, r = ctx.prove_sumcheck(ints(case["table"]), case["n"])
        assert r == ints(case["r"])
        assert all(right_aligned_equal(a, b) for a, b in zip(proof, ints(case["proof"])))


@pytest.mark.parametrize("n", [2, 3, 4, 5, 7, 9, 10, 11, 13])
def test_mle_random_tables_match_oracle(ctx, n):
    rng = random.Random(1000 + n)
    t = [rng.randrange(P) for _ in range(1 << n)]
    assert ctx.prove_sumcheck(t, n) == cdense.sumcheck_mle(t, n)


def test_mle_length_rule_edge_cases(ctx):
    for t, n in (([5] * 16, 4), ([i >> 1 for i in range(32)], 5), ([0] * 8, 3), ([1, 1, 2, 2], 2),
                 ([3, 4, 3, 4, 3, 4, 3, 4], 3), ([P - 1] * 64, 6)):
        assert ctx.prove_sumcheck(t, n) == dense.sumcheck_mle(t, n)


def test_mle_last_variable_dependence_on_streamed_tables(ctx):
    """The "does the table depend on x_n" bit of tables large enough for the streaming first pass (neighbour
    entries compared across lanes): independent of x_n, dependent through one single pair, constant."""
    n = 13
    pairs = [(i >> 1) * 7919 + 3 for i in range(1 << n)]
    one_pair = list(pairs)
    one_pair[5431] += 1
    last = list(pairs)
    last[-1] = 12345
    for t in (pairs, one_pair, last, [9] * (1 << n)):
        assert ctx.prove_sumcheck(t, n) == cdense.sumcheck_mle(t, n)


@pytest.mark.parametrize("n,seed", [(16, 0xC0FFEE + 1), (20, 0xC0FFEE + 2)])
def test_mle_baseline_sizes_match_oracle(ctx, n, seed):
    """configs 2/3 of BASELINE.json: 2^16 and 2^20 point tables, generated on the
    device by the same definition the oracle uses."""
    count = 1 << n
    d = ctx.alloc(count * 32)
    try:
        ctx.fill_table(d, count, seed)
        ctx.synchronize()
        host = ctx.download(d, (count, 4))
        ref_table = cdense.fill_table(count, seed)
        assert np.array_equal(host, ref_table)
        C, L, R = ctx.sumcheck_mle_batch_device(d, n, 1)
        c2, l2, r2 = cdense.sumcheck_mle_raw(ref_table, n)
        assert np.array_equal(C[0], c2) and np.array_equal(L[0], l2) and np.array_equal(R[0], r2)
        # the input table must be untouched
        assert np.array_equal(ctx.download(d, (count, 4)), ref_table)
    finally:
        ctx.free(d)


@pytest.mark.parametrize("n", [13, 14, 16, 17])
def test_mle_extreme_byte_patterns_match_oracle(ctx, n):
    """The fold pass multiplies on the matrix cores over signed bytes (mfma_fold.h): tables made of the byte
    patterns that sit on its sign and carry boundaries (0x00, 0x7f, 0x80, 0xff runs, p - 1, small values) must
    come out as the same field elements."""
    rng = random.Random(4000 + n)
    specials = [0, 1, P - 1, P - 2, (1 << 253) - 1, int.from_bytes(b"\x80" * 31 + b"\x20", "little"),
                int.from_bytes(b"\x7f" * 31 + b"\x2f", "little"), int.from_bytes(b"\xff" * 31 + b"\x2f", "little"),
                int.from_bytes(b"\x00\xff" * 15 + b"\x00\x30", "little"), 0x80, 0xff, 1 << 128]
    assert all(x < P for x in specials)
    for mode in ("all_max", "specials", "mixed"):
        if mode == "all_max":
            t = [P - 1] * (1 << n)
        elif mode == "specials":
            t = [specials[rng.randrange(len(specials))] for _ in range(1 << n)]
        else:
            t = [specials[rng.randrange(len(specials))] if rng.random() < 0.5 else rng.randrange(P) for _ in range(1 << n)]
        assert ctx.prove_sumcheck(t, n) == cdense.sumcheck_mle(t, n), mode


@pytest.mark.parametrize("env", [{"GKR_NO_MFMA_FOLD": "1"}, {"GKR_ROUNDS_PER_PASS": "1"}, {"GKR_ROUNDS_PER_PASS": "3"},
                                 {"GKR_ROUNDS_PER_PASS": "4"}, {"GKR_HASH_CHUNK": "16", "GKR_HOST_THREADS": "2"},
                                 {"GKR_NO_IFMA": "1"}, {"GKR_NO_IFMA": "1", "GKR_NO_ADX": "1"}, {"GKR_HOST_PASS_SCALAR": "1"}, {"GKR_PLAN_MAIN": "1", "GKR_FOLD_BLOCKS": "8192"},
                                 {"GKR_PASS_QUEUE_DEPTH": "1", "GKR_GROUP_SIZE": "3"}, {"GKR_PASS_QUEUE_DEPTH": "16", "GKR_GROUP_SIZE": "2"},
                                 {"GKR_NO_FUSED_REDUCE": "1"}, {"GKR_NO_FUSED_REDUCE": "1", "GKR_NO_MFMA_FOLD": "1"},
                                 {"GKR_DEVICE_HASH_PERCENT": "50"}, {"GKR_DEVICE_HASH_PERCENT": "90", "GKR_ROUNDS_PER_PASS": "3"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_fold_pass_variants_match_oracle(env):
    """Every schedule of the host-transcript sumcheck gives the same transcript: the v_mad_u64_u32 fold instead
    of the matrix-core one, 1 / 3 / 4 rounds per pass instead of 5, sixteen-lane and scalar host hashing, other
    block counts, many small groups with pass 0 queued one at a time or all at once, part of the batch hashed ON THE DEVICE
    (MiMC7 on eight lanes per element, the whole chain of passes without the host).  The knobs are read once per process,
    hence the child interpreter."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sizes = ((14, 20), (17, 3), (18, 2)) if "GKR_ROUNDS_PER_PASS" in env or "GKR_NO_MFMA_FOLD" in env else ((14, 20), (17, 3))
    if "GKR_DEVICE_HASH_PERCENT" in env:
        # a share of the batch hashed on the device (kernels_transcript.hip; batches of >= 64): one-block passes only,
        # every kind of pass, a ragged last wave of eight-lane groups, the long first pass
        sizes = ((6, 100), (10, 72), (14, 99), (17, 64), (20, 64))
    for n, batch in sizes:
        out = subprocess.run([sys.executable, os.path.join(here, "fold_variants_worker.py"), str(n), str(batch)],
                             env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("n,batch", [(6, 40), (9, 33), (10, 17), (11, 9), (12, 130), (15, 7), (17, 5), (18, 3), (19, 2), (21, 1)])
def test_mle_every_pass_schedule_matches_oracle(ctx, n, batch):
    """Table sizes on both sides of every switch in the pass schedule (one-block passes up to 2^9 entries, fold
    passes that stop at 2^12, 3/4/5 rounds per pass, one to four groups per batch), ragged batch sizes."""
    count = 1 << n
    d = ctx.alloc(batch * count * 32)
    try:
        for b in range(batch):
            ctx.fill_table(N.ctypes.c_void_p(d.value + b * count * 32), count, 31337 + 97 * n + b)
        C, L, R = ctx.sumcheck_mle_batch_device(d, n, batch)
    finally:
        ctx.free(d)
    for b in range(batch):
        c2, l2, r2 = cdense.sumcheck_mle_raw(cdense.fill_table(count, 31337 + 97 * n + b), n)
        assert np.array_equal(C[b], c2) and np.array_equal(L[b], l2) and np.array_equal(R[b], r2), (n, b)


@pytest.mark.parametrize("n", [14, 16, 20])
def test_lone_sumcheck_publishing_from_its_last_block_is_repeatable(ctx, n):
    """A lone sumcheck's fold passes publish from the last block to arrive (arrival counters, release / acquire across the
    XCDs' L2s, counters left zero for the next pass): the oracle's transcript, and the same bytes 500 times over."""
    count = 1 << n
    d = ctx.alloc(count * 32)
    try:
        ctx.fill_table(d, count, 4242 + n)
        want = cdense.sumcheck_mle_raw(cdense.fill_table(count, 4242 + n), n)
        for rep in range(500):
            C, L, R = ctx.sumcheck_mle_batch_device(d, n, 1)
            assert np.array_equal(C[0], want[0]) and np.array_equal(L[0], want[1]) and np.array_equal(R[0], want[2]), (n, rep)
    finally:
        ctx.free(d)


def test_mle_batch_is_independent_sumchecks(ctx):
    n, batch = 12, 5
    count = 1 << n
    d = ctx.alloc(batch * count * 32)
    try:
        for b in range(batch):
            ctx.fill_table(N.ctypes.c_void_p(d.value + b * count * 32), count, 77 + b)
        C, L, R = ctx.sumcheck_mle_batch_device(d, n, batch)
        for b in range(batch):
            c2, l2, r2 = cdense.sumcheck_mle_raw(cdense.fill_table(count, 77 + b), n)
            assert np.array_equal(C[b], c2) and np.array_equal(L[b], l2) and np.array_equal(R[b], r2)
    finally:
        ctx.free(d)


def test_mle_headline_workload_shape_matches_oracle():
    """What bench.py times, at a size the oracle can follow: 2^20-point tables in a batch of 256 (two scheduling
    groups of 128, the late stream on, all hashing threads racing for records) -- 24 tables spread over both groups,
    first and last included, byte for byte against the C oracle; every other table against the committed digests
    (tests/golden/bench_batch_hashes.json: the same seeds as the bench)."""
    import hashlib
    from gkr_amd import synth
    n, batch = 20, 256
    count = 1 << n
    with Context(0) as c:
        d = c.alloc(batch * count * 32)
        try:
            for b in range(batch):
                c.fill_table(N.ctypes.c_void_p(d.value + b * count * 32), count, synth.bench_table_seed(0, b))
            C, L, R = c.sumcheck_mle_batch_device(d, n, batch)
            C2, L2, R2 = c.sumcheck_mle_batch_device(d, n, batch, out=(np.zeros_like(C), np.zeros_like(L), np.zeros_like(R)))
        finally:
            c.free(d)
    assert np.array_equal(C, C2) and np.array_equal(L, L2) and np.array_equal(R, R2)   # a second step gives the same bytes
    picks = sorted({0, 1, 63, 64, 127, 128, 129, 191, 192, 254, 255} | {int(i * 255 / 12) for i in range(13)})
    assert len(picks) >= 16
    for b in picks:
        c2, l2, r2 = cdense.sumcheck_mle_raw(cdense.fill_table(count, synth.bench_table_seed(0, b)), n)
        assert np.array_equal(C[b], c2) and np.array_equal(L[b], l2) and np.array_equal(R[b], r2), b
    gold = synth.bench_batch_digests()
    assert gold is not None and gold["n"] == n
    for b in range(batch):
        h = hashlib.sha256(C[b].tobytes() + L[b].tobytes() + R[b].tobytes()).hexdigest()[:16]
        assert h == gold["rank0_tables"][b], b


@pytest.mark.parametrize("n", [24, 27, 30])
def test_mle_largest_tables(n):
    """Maximum sizes of the plain sumcheck: one table of 2^24 and 2^27 points (every byte against the oracle) and of
    2^30 points = 32 GiB, the ABI's limit (two runs agree; the verifier's sum / challenge relations over all 30 rounds).
    In a child process: tests/big_table_worker.py."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "big_table_worker.py"), str(n)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


def test_mle_verifier_relation_at_full_size(ctx):
    """size-independent property at 2^20: g_j(0)+g_j(1) = g_{j-1}(r_{j-1}), r_j = MiMC(g_j)."""
    n = 20
    count = 1 << n
    d = ctx.alloc(count * 32)
    try:
        ctx.fill_table(d, count, 4242)
        C, L, R = ctx.sumcheck_mle_batch_device(d, n, 1)
    finally:
        ctx.free(d)
    claim = None
    for j in range(n):
        c1, c0 = cdense.from_limbs(C[0, j])
        r = cdense.from_limbs(R[0, j])[0]
        if claim is not None:
            assert (2 * c0 + c1) % P == claim
        vec = [c1, c0][2 - int(L[0, j]):]
        assert gkr_amd.multi_hash(vec) == r
        claim = (c1 * r + c0) % P


# ---------------------------------------------------------------- layer pieces

@pytest.mark.parametrize("seed", range(4))
def test_predicates_and_layer_eval_match_oracle(ctx, seed):
    rng = random.Random(2000 + seed)
    k_i, k = rng.randint(0, 9), rng.randint(1, 4)
    g = 1 << k_i
    lay = Layer(k_i, [rng.randint(0, 1) for _ in range(g)], [rng.randrange(1 << k) for _ in range(g)],
                [rng.randrange(1 << k) for _ in range(g)])
    z = [rng.randrange(P) for _ in range(k_i)]
    A, M = ctx.predicate_tables(lay, k, z)
    a, m = cdense.predicate_tables(k_i, k, lay.gate_type, lay.left, lay.right, z)
    assert np.array_equal(A, a) and np.array_equal(M, m)
    prev = [rng.randrange(P) for _ in range(1 << k)]
    out = ctx.layer_eval(lay, prev)
    assert cdense.from_limbs(out) == dense.layer_eval(lay.gate_type, lay.left, lay.right, prev)


def test_predicates_with_heavy_collisions(ctx):
    """many gates on one (left, right) cell: the widened-atomic scatter must stay exact."""
    k_i, k = 12, 1
    g = 1 << k_i
    lay = Layer(k_i, [i & 1 for i in range(g)], [0] * g, [1] * g)
    z = [random.Random(5).randrange(P) for _ in range(k_i)]
    A, M = ctx.predicate_tables(lay, k, z)
    a, m = cdense.predicate_tables(k_i, k, lay.gate_type, lay.left, lay.right, z)
    assert np.array_equal(A, a) and np.array_equal(M, m)


@pytest.mark.parametrize("seed", range(8))
def test_layer_sumcheck_matches_oracle(ctx, seed):
    rng = random.Random(3000 + seed)
    k_i, k = rng.randint(0, 8), rng.randint(1, 5)
    g = 1 << k_i
    gt = [rng.randint(0, 1) for _ in range(g)]
    if seed % 4 == 0:
        gt = [0] * g
    if seed % 4 == 1:
        gt = [1] * g
    lay = Layer(k_i, gt, [rng.randrange(1 << k) for _ in range(g)], [rng.randrange(1 << k) for _ in range(g)])
    z = [rng.randrange(P) for _ in range(k_i)]
    mode = seed % 3
    if mode == 0:
        w = [rng.randrange(P) for _ in range(1 << k)]
    elif mode == 1:
        w = [(i >> (k - 1)) + 1 for i in range(1 << k)]     # depends on x1 only: short rounds
    else:
        w = [rng.randrange(2) for _ in range(1 << k)]
    assert ctx.prove_sumcheck_opt(lay, k, z, w) == cdense.sumcheck_layer(k_i, k, gt, lay.left, lay.right, z, w)


@pytest.mark.parametrize("env", [{}, {"GKR_NO_CIRCUIT_CACHE": "1"}, {"GKR_HOST_THREADS": "2", "GKR_HASH_CHUNK": "16"}, {"GKR_NO_IFMA": "1"},
                                 {"GKR_NO_IFMA": "1", "GKR_NO_ADX": "1"}, {"GKR_LINE_STEPWISE": "1"}, {"GKR_NO_FUSED_PUBLISH": "1"},
                                 {"GKR_GATE_SORT_GLOBAL": "1"}, {"GKR_GATE_GROUPS_MIN_K": "2"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()) or "default")
def test_layer_path_variants_match_oracle(env):
    """The layer sumcheck with the host transcript: linear time over (W, U, V) tables of 2^k entries summed straight from
    the gate lists, product passes of up to three rounds per device round trip -- under every switch that still reaches
    it (csrc/options.h), single layers and a batch of proofs, IFMA-lane and scalar host passes.  A child process: the
    environment seeds a context's options when it is created, and the process switches are read once."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "layer_variants_worker.py")], env=dict(os.environ, **env),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


def test_options_are_per_context():
    """gkr_ctx_set_option: two contexts of ONE process with different switches give the same transcripts (the options are
    read per context, not from process-wide statics), an unknown name is an error, and the device transcript's two b-phase
    forms (layer_no_fused) agree."""
    from gkr_amd import Context, GkrError
    rng = random.Random(4242)
    n = 13
    table = [rng.randrange(P) for _ in range(1 << n)]
    k_i, k = 9, 5
    lay = Layer(k_i, [rng.randint(0, 1) for _ in range(1 << k_i)], [rng.randrange(1 << k) for _ in range(1 << k_i)],
                [rng.randrange(1 << k) for _ in range(1 << k_i)])
    z = [rng.randrange(P) for _ in range(k_i)]
    w = [rng.randrange(P) for _ in range(1 << k)]
    want_mle = cdense.sumcheck_mle(table, n)
    want_layer = cdense.sumcheck_layer(k_i, k, lay.gate_type, lay.left, lay.right, z, w)
    with Context(0) as a, Context(0) as b, Context(0) as c:
        b.set_option("no_mfma_fold", 1)
        b.set_option("GKR_ROUNDS_PER_PASS", 2)          # (the variable's name is accepted too)
        b.set_option("gate_groups_min_k", 1)
        b.set_option("no_fused_publish", 1)
        c.set_option("mle_per_round", 1)
        c.set_option("line_stepwise", 1)
        b.set_option("host_tail_log2", -1)             # (no host tail: the plain sumcheck's last fold pass and the layers' last product passes on the device)
        assert a.get_option("rounds_per_pass") == 0 and b.get_option("rounds_per_pass") == 2 and c.get_option("mle_per_round") == 1
        for ctx_ in (a, b, c, a):
            assert ctx_.prove_sumcheck(table, n) == want_mle
            assert ctx_.prove_sumcheck_opt(lay, k, z, w) == want_layer
        with pytest.raises(GkrError):
            a.set_option("no_such_option", 1)
        a.set_transcript(N.GKR_TRANSCRIPT_DEVICE)
        assert a.prove_sumcheck_opt(lay, k, z, w) == want_layer
        a.set_option("layer_no_fused", 1)
        assert a.prove_sumcheck_opt(lay, k, z, w) == want_layer


def test_layer_sumcheck_length_rule_edges(ctx):
    gates = Layer(2, [0, 1, 0, 1], [0, 1, 2, 3], [3, 2, 1, 0])
    for w in ([1, 2, 1, 2], [1, 1, 2, 2], [7, 7, 7, 7], [0, 0, 0, 0]):
        assert ctx.prove_sumcheck_opt(gates, 2, [11, 13], w) == dense.sumcheck_layer(2, 2, gates.gate_type, gates.left,
                                                                                     gates.right, [11, 13], w)
    one = Layer(1, [0, 1], [0, 1], [1, 1])
    p, _ = ctx.prove_sumcheck_opt(one, 1, [7], [3, 3])
    assert p[0] == [99, P - 36]


def test_layer_sumcheck_wide(ctx):
    """k = 7 (2^14-point hypercube), 2^14 gates."""
    rng = random.Random(99)
    k_i, k = 14, 7
    g = 1 << k_i
    lay = Layer(k_i, [rng.randint(0, 1) for _ in range(g)], [rng.randrange(1 << k) for _ in range(g)],
                [rng.randrange(1 << k) for _ in range(g)])
    z = [rng.randrange(P) for _ in range(k_i)]
    w = [rng.randrange(P) for _ in range(1 << k)]
    assert ctx.prove_sumcheck_opt(lay, k, z, w) == cdense.sumcheck_layer(k_i, k, lay.gate_type, lay.left, lay.right, z, w)


def test_layer_sumcheck_golden_layers(ctx, gkr_cases):
    """every layer of every fixture, driven with the fixture's own z[i]
    (covers python/gkr.py's random z[0])."""
    for case in gkr_cases:
        vals = ints(case["values"])
        for i, (gt, l, r) in enumerate(layers_of(case)):
            k_i, k = case["k"][i], case["k"][i + 1]
            if not all(dense.depends_on(vals[i + 1], k)):
                continue
            proof, rs = ctx.prove_sumcheck_opt(Layer(k_i, gt, l, r), k, ints(case["z"][i]), vals[i + 1])
            assert rs == ints(case["sumcheck_r"][i])
            assert all(right_aligned_equal(a, b) for a, b in zip(proof, ints(case["sumcheck_proofs"][i])))


# ---------------------------------------------------------------- full proofs

def test_prove_matches_reference_python_fixtures(ctx, gkr_cases):
    n_checked = 0
    for case in gkr_cases:
        vals = ints(case["values"])
        generic = all(all(dense.depends_on(v, k)) for v, k in zip(vals[1:], case["k"][1:]))
        if not generic or any(ints(case["z0"])):
            continue        # the Rust prover fixes z[0] = 0 (prover.rs:16-21)
        pr = ctx.prove(_circuit(layers_of(case), len(case["inputs"])), ints(case["inputs"]))
        assert pr.k == case["k"] and pr.depth == len(case["k"])
        assert pr.sumcheck_r == ints(case["sumcheck_r"])
        assert pr.z == ints(case["z"]) and pr.r == ints(case["r"])
        for lay in range(len(pr.q)):
            assert right_aligned_equal(pr.q[lay], ints(case["q"][lay]))
            assert all(right_aligned_equal(a, b)
                       for a, b in zip(pr.sumcheck_proofs[lay], ints(case["sumcheck_proofs"][lay])))
        if case["k"][0] > 0:
            assert terms_as_set(pr.d) == terms_as_set(ints(case["D"]))
        assert terms_as_set(pr.input_func) == terms_as_set(ints(case["input_func"]))
        n_checked += 1
    assert n_checked >= 5


@pytest.mark.parametrize("seed", range(3))
def test_prove_random_circuits_match_oracle(ctx, seed):
    rng = random.Random(4000 + seed)
    ks = [rng.randint(0, 5), rng.randint(1, 5), rng.randint(1, 5), rng.randint(1, 4)]
    layers = []
    for i in range(len(ks) - 1):
        g, n = 1 << ks[i], 1 << ks[i + 1]
        layers.append(([rng.randint(0, 1) for _ in range(g)], [rng.randrange(n) for _ in range(g)],
                       [rng.randrange(n) for _ in range(g)]))
    inputs = [rng.randrange(P) for _ in range(1 << ks[-1])]
    pr = ctx.prove(_circuit(layers, len(inputs)), inputs)
    ref = cdense.prove(layers, inputs)
    assert pr.sumcheck_proofs == ref["sumcheck_proofs"] and pr.sumcheck_r == ref["sumcheck_r"]
    assert pr.q == ref["q"] and pr.z == ref["z"] and pr.r == ref["r"] and pr.k == ref["k"]
    full = dense.prove(layers, inputs) if max(ks) <= 3 else None
    if full:
        assert terms_as_set(pr.d) == terms_as_set(full["d"])
    assert terms_as_set(pr.input_func) == terms_as_set(dense.monomial_terms(inputs, ks[-1]))


def test_prove_wide_layers_match_oracle(ctx):
    """A circuit with layers of 2^9 .. 2^11 values: the line restriction q in its wide-layer form (one launch per
    variable), product passes that span several blocks per proof; one proof and a batch of three.  Inputs with few
    distinct values give W layers whose monomial degree -- the length of q -- is below k."""
    rng = random.Random(4100)
    ks = [2, 10, 11, 9]
    layers = []
    for i in range(len(ks) - 1):
        g, n = 1 << ks[i], 1 << ks[i + 1]
        layers.append(([rng.randint(0, 1) for _ in range(g)], [rng.randrange(n) for _ in range(g)],
                       [rng.randrange(n) for _ in range(g)]))
    witnesses = [[rng.randrange(P) for _ in range(1 << ks[-1])], [(i >> 7) + 3 for i in range(1 << ks[-1])],
                 [rng.randrange(2) for _ in range(1 << ks[-1])]]
    refs = [cdense.prove(layers, w) for w in witnesses]
    circuit = _circuit(layers, 1 << ks[-1])
    for got in ([ctx.prove(circuit, witnesses[0])], ctx.prove_batch(circuit, witnesses)):
        for pr, ref in zip(got, refs):
            assert pr.sumcheck_proofs == ref["sumcheck_proofs"] and pr.sumcheck_r == ref["sumcheck_r"]
            assert pr.q == ref["q"] and pr.z == ref["z"] and pr.r == ref["r"] and pr.k == ref["k"]


@pytest.mark.parametrize("tail_log2,max_batch", [(-1, 0), (0, 0), (3, 0), (9, 0), (12, 0), (6, 2), (8, 64)])
def test_host_tail_of_the_product_passes(tail_log2, max_batch):
    """host_tail_log2 / host_tail_max_batch: a phase's passes over small tables run on the host (capi_layer.hip, host_tail_pass)
    from tables the last device pass leaves in pinned memory -- from the phase's FIRST pass on when the layer is that small, never
    (-1), for a batch above the limit not at all.  One proof, a batch of three (constant and 0/1 witnesses: short round
    vectors) and a batch of nine against the checker, layers of 2^2 .. 2^11 values; every proof passes through both phases'
    hand-over (the c-phase's set-up reads what is left of W from the device)."""
    rng = random.Random(5150 + tail_log2)
    ks = [2, 10, 11, 7, 4, 9]
    layers = []
    for i in range(len(ks) - 1):
        g, n = 1 << ks[i], 1 << ks[i + 1]
        layers.append(([rng.randint(0, 1) for _ in range(g)], [rng.randrange(n) for _ in range(g)],
                       [rng.randrange(n) for _ in range(g)]))
    witnesses = [[rng.randrange(P) for _ in range(1 << ks[-1])], [(i >> 7) + 3 for i in range(1 << ks[-1])],
                 [rng.randrange(2) for _ in range(1 << ks[-1])]] + [[rng.randrange(P) for _ in range(1 << ks[-1])] for _ in range(6)]
    refs = [cdense.prove(layers, w) for w in witnesses]
    circuit = _circuit(layers, 1 << ks[-1])
    with Context(0) as c:
        c.set_option("host_tail_log2", tail_log2)
        c.set_option("host_tail_max_batch", max_batch)
        for got, want in (([c.prove(circuit, witnesses[0])], refs[:1]), (c.prove_batch(circuit, witnesses[:3]), refs[:3]),
                          (c.prove_batch(circuit, witnesses), refs)):
            for pr, ref in zip(got, want):
                assert pr.sumcheck_proofs == ref["sumcheck_proofs"] and pr.sumcheck_r == ref["sumcheck_r"]
                assert pr.q == ref["q"] and pr.z == ref["z"] and pr.r == ref["r"] and pr.k == ref["k"]


def test_prove_batch_equals_single_proofs_and_oracle(ctx):
    """gkr_prove_batch: proofs of one circuit for many witnesses advanced together (host transcript), or layer by
    layer one after the other with no host in the loop (device transcript)."""
    rng = random.Random(4242)
    ks = [3, 4, 5, 4]
    layers = []
    for i in range(len(ks) - 1):
        g, n = 1 << ks[i], 1 << ks[i + 1]
        layers.append(([rng.randint(0, 1) for _ in range(g)], [rng.randrange(n) for _ in range(g)],
                       [rng.randrange(n) for _ in range(g)]))
    circ = _circuit(layers, 1 << ks[-1])
    B = 19   # not a multiple of the eight hash lanes
    inputs = [[rng.randrange(P) for _ in range(1 << ks[-1])] for _ in range(B)]
    inputs[3] = [7] * (1 << ks[-1])                       # constant inputs: short round vectors
    inputs[5] = [i & 1 for i in range(1 << ks[-1])]       # depends on the last variable only
    got = ctx.prove_batch(circ, inputs)
    for b in (0, 3, 5, 11, 18):
        ref = cdense.prove(layers, inputs[b])
        pr = got[b]
        assert pr.sumcheck_proofs == ref["sumcheck_proofs"] and pr.sumcheck_r == ref["sumcheck_r"], b
        assert pr.q == ref["q"] and pr.z == ref["z"] and pr.r == ref["r"] and pr.k == ref["k"], b
    one = ctx.prove(circ, inputs[7])
    assert one == got[7]


def test_product_proofs_pass_the_verifier(ctx):
    """every proof the GPU prover makes satisfies the reference verifier's relations (python/gkr.py:202-231)"""
    from gkr_amd import verify
    rng = random.Random(777)
    ks = [2, 4, 5, 5]
    layers = []
    for i in range(len(ks) - 1):
        g, n = 1 << ks[i], 1 << ks[i + 1]
        layers.append(([rng.randint(0, 1) for _ in range(g)], [rng.randrange(n) for _ in range(g)],
                       [rng.randrange(n) for _ in range(g)]))
    circ = _circuit(layers, 1 << ks[-1])
    pr = ctx.prove(circ, [rng.randrange(P) for _ in range(1 << ks[-1])])
    assert verify(pr, circ)
    pr.sumcheck_proofs[1][2][-1] = (pr.sumcheck_proofs[1][2][-1] + 1) % P
    assert not verify(pr, circ)


def test_proofs_format_as_verifier_circom_inputs(ctx, gkr_cases):
    """The artefact after the path (aggregator.rs:92-213, file_utils.rs:49-67) from the GPU prover's own proofs."""
    from gkr_amd.aggregate import aggregated_input, circom_input, circom_meta
    from helpers import canon_circom, expected_circom_input, expected_circom_meta
    proofs = []
    for case in gkr_cases[:4]:
        layers = layers_of(case)
        p = ctx.prove(_circuit(layers, len(case["inputs"])), ints(case["inputs"]))
        assert circom_meta(p) == expected_circom_meta(p)
        assert canon_circom(circom_input(p, len(proofs))) == canon_circom(expected_circom_input(p, len(proofs)))
        proofs.append(p)
    merged = aggregated_input({"a": "1"}, proofs)
    assert len(merged) == 1 + 7 * len(proofs)


def test_prove_zero_output_check(ctx):
    # convert.rs:838: the reference asserts output 0 == 0 on a satisfying witness
    circ = _circuit([([0, 1], [0, 1], [1, 1])], 2)
    with pytest.raises(GkrError):
        ctx.prove(circ, [3, 4], require_zero_output=True)
    pr = ctx.prove(circ, [0, 0], require_zero_output=True)
    assert pr.d == [] and pr.sumcheck_proofs[0][0][-1] == 0


# ---------------------------------------------------------------- error behaviour

def test_errors_are_statuses_not_aborts(ctx):
    lay = Layer(1, [0, 1], [0, 1], [1, 1])
    with pytest.raises(GkrError) as e:
        ctx.prove_sumcheck_opt(lay, 0, [1], [5])                  # v == 0 (sumcheck.rs:49 underflow)
    assert e.value.status == N.GKR_ERR_DEGENERATE
    with pytest.raises(GkrError) as e:
        ctx.prove_sumcheck_opt(Layer(1, [0, 1], [0, 2], [1, 1]), 1, [1], [5, 6])   # operand out of range
    assert e.value.status == N.GKR_ERR_INVALID
    bad = np.zeros((4, 4), dtype=np.uint64)
    bad[2, :] = np.uint64(0xFFFFFFFFFFFFFFFF)                       # >= r
    with pytest.raises(GkrError) as e:
        ctx.sumcheck_mle_raw(bad, 2)
    assert e.value.status == N.GKR_ERR_NON_CANONICAL
    with pytest.raises(GkrError):
        ctx.sumcheck_mle_raw(np.zeros((2, 4), dtype=np.uint64), 1)  # n < 2
