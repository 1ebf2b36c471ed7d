"""GPU parity of the multi-GPU algorithm with logical ranks on one device: the hypercube is cut
into P trailing-variable shards, every shard is a library session, the per-round reduce is done
in-process.  Results must equal the unsharded oracle bit for bit."""

import ctypes
import random

import numpy as np
import pytest

from gkr_amd import Context, Layer, parallel
from gkr_amd.field import to_limbs
from oracle import cdense, dense
from oracle.field import P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def _layer(rng, k_i, k):
    g = 1 << k_i
    return Layer(k_i, [rng.randint(0, 1) for _ in range(g)], [rng.randrange(1 << k) for _ in range(g)],
                 [rng.randrange(1 << k) for _ in range(g)])


@pytest.mark.parametrize("nshards", [1, 2, 4, 8])
def test_layer_sumcheck_logical_ranks(ctx, nshards):
    rng = random.Random(500 + nshards)
    for k_i, k in ((4, 3), (6, 4), (0, 3)):
        lay = _layer(rng, k_i, k)
        z = [rng.randrange(P) for _ in range(k_i)]
        for w in ([rng.randrange(P) for _ in range(1 << k)], [(i >> (k - 1)) + 1 for i in range(1 << k)],
                  [(i & 1) + 5 for i in range(1 << k)]):
            got = parallel.prove_sumcheck_opt_logical(ctx, lay, k, z, w, nshards)
            assert got == cdense.sumcheck_layer(k_i, k, lay.gate_type, lay.left, lay.right, z, w), (nshards, k_i, k)


@pytest.mark.parametrize("nshards", [1, 2, 3, 4, 8])
def test_layer_sumcheck_gate_sharded_logical_ranks(nshards):
    """The gate-sharded form (gkr_sumcheck_layer_sharded): any partition of the gates, also a world that is not a
    power of two; every rank ends with the unsharded oracle's transcript, short round vectors included."""
    rng = random.Random(1500 + nshards)
    for k_i, k in ((4, 3), (7, 4), (0, 3), (2, 5), (10, 6)):
        lay = _layer(rng, k_i, k)
        z = [rng.randrange(P) for _ in range(k_i)]
        for w in ([rng.randrange(P) for _ in range(1 << k)], [(i >> (k - 1)) + 1 for i in range(1 << k)],
                  [(i & 1) + 5 for i in range(1 << k)]):
            want = cdense.sumcheck_layer(k_i, k, lay.gate_type, lay.left, lay.right, z, w)
            got = parallel.prove_sumcheck_opt_logical_gates(0, lay, k, z, w, nshards)
            assert len(got) == nshards and all(g == want for g in got), (nshards, k_i, k)


@pytest.mark.parametrize("nshards", [1, 2, 3, 8])
def test_layer_sumcheck_gate_sharded_device_exchange(nshards):
    """The same split with the DEVICE exchange (gkr_resident_layer_sumcheck_dev): partial tables widened, summed and
    narrowed on the GPU, no host copy; every rank's transcript equals the unsharded oracle's."""
    rng = random.Random(1700 + nshards)
    for k_i, k in ((7, 4), (0, 3), (10, 6), (16, 8)):
        lay = _layer(rng, k_i, k)
        z = to_limbs([rng.randrange(P) for _ in range(k_i)])
        w = to_limbs([rng.randrange(P) for _ in range(1 << k)] if k_i != 10 else [(i & 1) + 5 for i in range(1 << k)])
        want = cdense.sumcheck_layer_raw(k_i, k, lay.gate_type, lay.left, lay.right, z, w)
        got = parallel.prove_sumcheck_opt_logical_gates_dev(0, lay, k, z, w, nshards)
        assert len(got) == nshards
        for C, L, R in got:
            assert np.array_equal(C, want[0]) and np.array_equal(L, want[1]) and np.array_equal(R, want[2]), (nshards, k_i, k)


def test_device_exchange_bad_gate_fails_every_rank():
    from gkr_amd import GkrError
    rng = random.Random(1601)
    lay = _layer(rng, 6, 4)
    lay.right[-1] = 1 << 4
    with pytest.raises(GkrError):
        parallel.prove_sumcheck_opt_logical_gates_dev(0, lay, 4, [rng.randrange(P) for _ in range(6)],
                                                      [rng.randrange(P) for _ in range(16)], 4)


def test_device_exchange_argument_errors_are_statuses(ctx):
    """gkr_resident_layer_sumcheck_dev: a buffer smaller than gkr_exchange_limbs(k_next), a null hook, and a hook that
    reports failure come back as error statuses (the hook's failure after both kernels of the exchange were queued)."""
    import torch
    from gkr_amd import GkrError
    from gkr_amd import _native as N
    rng = random.Random(1702)
    k_i, k = 6, 4
    lay = _layer(rng, k_i, k)
    gt, l, r = lay.arrays()
    z = to_limbs([rng.randrange(P) for _ in range(k_i)])
    w = to_limbs([rng.randrange(P) for _ in range(1 << k)])
    need = int(N.lib().gkr_exchange_limbs(ctypes.c_int(k)))
    assert need == ((2 << k) + 1) * 8 and N.lib().gkr_exchange_limbs(ctypes.c_int(99)) == 0
    buf = torch.zeros(need, dtype=torch.int64, device="cuda:0")
    calls = []

    def failing(_user, count, stream):
        calls.append(count)
        return 7
    gates = parallel.ResidentGates(ctx, k_i, 0, gt, l, r)
    try:
        for fn, capacity in ((N.ALLREDUCE_DEV_FN(failing), need - 1), (N.ALLREDUCE_DEV_FN(0), need), (N.ALLREDUCE_DEV_FN(failing), need)):
            ex = parallel.DeviceExchange.__new__(parallel.DeviceExchange)
            ex.errors, ex.calls, ex._buf, ex._fn = [], 0, buf, fn
            ex.struct = N.ExchangeDev(fn, None, buf.data_ptr(), capacity)
            with pytest.raises(GkrError):
                gates.sumcheck_raw(k, z, w, ex)
        assert calls == [need]      # only the last case reaches the hook
        # and the layer still proves afterwards (whole layer, no exchange)
        want = cdense.sumcheck_layer_raw(k_i, k, gt, l, r, z, w)
        got = gates.sumcheck_raw(k, z, w)
        assert all(np.array_equal(a, b) for a, b in zip(got, want))
    finally:
        gates.close()


def test_device_exchange_over_rccl_single_rank():
    """RCCL itself in the loop: a process group with backend nccl (world of one rank -- one MI355X is what this box
    has), the layer's two exchanges as all-reduces queued on the library's stream through torch.distributed, and the plain
    sumcheck's per-round all-reduce of the trailing-variable form; transcripts = the oracle's."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(here, "rccl_exchange_worker.py")], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


def test_native_rccl_exchange_single_rank():
    """The library's OWN collective (gkr_exchange_rccl_*: librccl loaded by the library, ncclAllReduce(ncclInt64, ncclSum)
    queued by its own hook on its own stream): the gate-sharded layer sumcheck and the split plain sumcheck through it, in
    a process that never imports torch."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(here, "rccl_native_worker.py")], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


def test_gate_sharded_bad_gate_fails_every_rank():
    """A bad gate on ONE rank: every rank must come back with an error (none may stay inside the collective)."""
    from gkr_amd import GkrError
    rng = random.Random(1600)
    lay = _layer(rng, 6, 4)
    lay.left[-1] = 1 << 4   # operand out of range, in the last rank's share
    with pytest.raises(GkrError):
        parallel.prove_sumcheck_opt_logical_gates(0, lay, 4, [rng.randrange(P) for _ in range(6)],
                                                  [rng.randrange(P) for _ in range(16)], 4)


def _decode(C, L, R, k):
    from gkr_amd.field import from_limbs
    proof = [from_limbs(C[j])[3 - int(L[j]):] for j in range(2 * k)]
    return proof, from_limbs(R)


def test_resident_layer_many_sumchecks_and_bad_gates(ctx):
    """gkr_resident_layer_*: the gates and their sorted lists stay on the device; several (z, W) on one layer -- widths
    with passes of one and of several blocks per proof -- equal the oracle, a second next-layer width gets its own lists,
    a gate out of range fails the first sumcheck (and the one after it) with GKR_ERR_INVALID."""
    from gkr_amd import GkrError
    from gkr_amd.field import as_limbs
    rng = random.Random(1700)
    for k_i, k in ((9, 5), (12, 9), (3, 1), (0, 2)):
        lay = _layer(rng, k_i, k)
        gates = parallel.ResidentGates(ctx, k_i, 0, *lay.arrays())
        try:
            for trial in range(3):
                z = [rng.randrange(P) for _ in range(k_i)]
                w = [rng.randrange(P) for _ in range(1 << k)] if trial != 1 else [(i >> (k - 1)) + 1 for i in range(1 << k)]
                C, L, R = gates.sumcheck_raw(k, as_limbs(z) if k_i else np.zeros((0, 4), dtype=np.uint64), as_limbs(w))
                assert _decode(C, L, R, k) == cdense.sumcheck_layer(k_i, k, lay.gate_type, lay.left, lay.right, z, w), (k_i, k, trial)
        finally:
            gates.close()
    lay = _layer(rng, 8, 4)
    lay.right[17] = 16
    gates = parallel.ResidentGates(ctx, 8, 0, *lay.arrays())
    try:
        z, w = [rng.randrange(P) for _ in range(8)], [rng.randrange(P) for _ in range(16)]
        for _ in range(2):
            with pytest.raises(GkrError):
                gates.sumcheck_raw(4, as_limbs(z), as_limbs(w))
    finally:
        gates.close()


def test_layer_sumcheck_logical_ranks_wide(ctx):
    rng = random.Random(77)
    k_i, k = 12, 6
    lay = _layer(rng, k_i, k)
    z = [rng.randrange(P) for _ in range(k_i)]
    w = [rng.randrange(P) for _ in range(1 << k)]
    ref = cdense.sumcheck_layer(k_i, k, lay.gate_type, lay.left, lay.right, z, w)
    assert parallel.prove_sumcheck_opt_logical(ctx, lay, k, z, w, 8) == ref
    assert ctx.prove_sumcheck_opt(lay, k, z, w) == ref


def test_layer_session_as_single_rank_world(ctx):
    rng = random.Random(78)
    k_i, k = 5, 4
    lay = _layer(rng, k_i, k)
    z = [rng.randrange(P) for _ in range(k_i)]
    w = [rng.randrange(P) for _ in range(1 << k)]
    s = parallel.LayerSession.open(ctx, lay, k, z, w)
    got = parallel.prove_sumcheck_opt_distributed(s, parallel.SingleProcess(), k, s.dep(k), None)
    s.close()
    assert got == dense.sumcheck_layer(k_i, k, lay.gate_type, lay.left, lay.right, z, w)


def _upload_shards(ctx, table, nshards):
    ptrs = []
    for p in range(nshards):
        part = to_limbs(table[p::nshards])
        d = ctx.alloc(part.nbytes)
        ctx.upload(d, part)
        ptrs.append(d)
    return ptrs


@pytest.mark.parametrize("nshards", [1, 2, 4, 8])
def test_mle_sumcheck_logical_ranks(ctx, nshards):
    rng = random.Random(900 + nshards)
    n = 10
    tables = [[rng.randrange(P) for _ in range(1 << n)], [5] * (1 << n), [i >> 1 for i in range(1 << n)],
              [rng.randrange(2) for _ in range(1 << n)]]
    for t in tables:
        ptrs = _upload_shards(ctx, t, nshards)
        try:
            assert parallel.prove_sumcheck_logical(ctx, ptrs, n) == cdense.sumcheck_mle(t, n)
        finally:
            for d in ptrs:
                ctx.free(d)


def test_tables_differ(ctx):
    a = to_limbs(list(range(1000)))
    b = a.copy()
    da, db = ctx.alloc(a.nbytes), ctx.alloc(b.nbytes)
    try:
        ctx.upload(da, a)
        ctx.upload(db, b)
        assert parallel.tables_differ(ctx, da, db, 1000) is False
        b[777, 2] = np.uint64(1)
        ctx.upload(db, b)
        assert parallel.tables_differ(ctx, da, db, 1000) is True
    finally:
        ctx.free(da)
        ctx.free(db)
