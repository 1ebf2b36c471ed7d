"""GPU parity of ONE plain sumcheck split over ranks on the multi-round schedule (gkr_sumcheck_mle_sharded_dev): the
reduce over the hypercube of prove_sumcheck (rust/src/gkr/sumcheck.rs:158-214, the rayon reduce of :62) as one all-reduce
per pass of up to five rounds.  Logical ranks on the one visible GPU (a thread and a context per rank, the all-reduce an
in-process device sum), RCCL with one rank (tests/test_gpu_sharded.py's worker), and two processes over gloo sharing the
GPU.  Every rank's transcript must equal the unsharded C oracle's byte for byte."""

import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from gkr_amd import Context, GkrError, parallel
from oracle import cdense

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _check(got, tables, n):
    for C, L, R, _ in got:
        for b in range(tables.shape[0]):
            want = cdense.sumcheck_mle_raw(tables[b], n)
            assert np.array_equal(C[b], want[0]) and np.array_equal(L[b], want[1]) and np.array_equal(R[b], want[2]), (n, b)


@pytest.mark.parametrize("nshards", [1, 2, 4, 8])
@pytest.mark.parametrize("n", [20, 24])
def test_split_sumcheck_logical_ranks_baseline_sizes(n, nshards):
    tables = cdense.fill_table(1 << n, 0xC0FFEE + 2)[None]        # n = 20: the table of BASELINE configs[2]
    got = parallel.prove_sumcheck_logical_dev(0, tables, n, nshards)
    _check(got, tables, n)
    # exchanges per sumcheck: one per pass of up to five rounds + the gather -- not one per round
    assert all(g[3] == got[0][3] for g in got) and got[0][3] <= 6, [g[3] for g in got]
    if n == 20 and nshards == 8:
        assert got[0][3] == 4


@pytest.mark.parametrize("nshards", [1, 2, 4, 8, 16])
def test_split_sumcheck_small_and_edge_tables(nshards):
    lp = nshards.bit_length() - 1
    rng = np.random.default_rng(60 + nshards)
    for n in sorted({lp + 1, lp + 2, lp + 6, lp + 7, lp + 9, 13}):
        if n < 2:
            continue
        rnd = cdense.fill_table(1 << n, 900 + n)
        half = cdense.fill_table(1 << (n - 1), 901 + n)
        tables = np.stack([rnd,
                           np.repeat(half, 2, axis=0),                       # does not depend on the last variable
                           np.repeat(rnd[:1], 1 << n, axis=0),               # constant
                           np.concatenate([half, half])])                    # does not depend on the first variable
        got = parallel.prove_sumcheck_logical_dev(0, tables, n, nshards)
        _check(got, tables, n)
    del rng


def test_split_sumcheck_limits_and_errors():
    with Context(0) as ctx:
        ex = parallel.NoExchange(ctx, parallel.exchange_limbs_mle(10, 0, 1))
        d = ctx.alloc(32 << 10)
        try:
            with pytest.raises(GkrError):
                parallel.sumcheck_mle_sharded_raw(ctx, d, 10, 2, 4, ex)               # shard 4 of 4
            with pytest.raises(GkrError):
                parallel.sumcheck_mle_sharded_raw(ctx, d, 10, 0, 0, ex, batch=64)     # exchange buffer too small for the batch
            with pytest.raises(GkrError):
                parallel.sumcheck_mle_sharded_raw(ctx, d, 1, 0, 0, ex)                # n < 2
            assert parallel.exchange_limbs_mle(40, 2, 1) == 0                         # n - log2 P beyond GKR_MAX_MLE_N
            # one rank, no exchange partner: the entry point is the plain sumcheck
            t = cdense.fill_table(1 << 10, 5)
            ctx.upload(d, t)
            C, L, R, nx = parallel.sumcheck_mle_sharded_raw(ctx, d, 10, 0, 0, ex)
            want = cdense.sumcheck_mle_raw(t, 10)
            assert np.array_equal(C[0], want[0]) and np.array_equal(L[0], want[1]) and np.array_equal(R[0], want[2]) and nx == ex.calls
        finally:
            ctx.free(d)
            ex.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_split_sumcheck_two_processes_over_gloo_one_gpu(tmp_path):
    """Two ranks as two processes (torch.distributed.run, backend gloo, both on the one GPU): each uploads its shard and
    calls gkr_sumcheck_mle_sharded_dev with the staged exchange; both must return the oracle's transcript."""
    env = dict(os.environ, GKR_TEST_OUT=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(HERE, "mle_split_worker.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    outs = [json.load(open(tmp_path / ("split_rank%d.json" % r))) for r in range(2)]
    assert all(o["ok"] and o["world"] == 2 and o["backend"] == "gloo" for o in outs), outs
    assert outs[0]["exchanges"] == outs[1]["exchanges"] and outs[0]["digest"] == outs[1]["digest"]
