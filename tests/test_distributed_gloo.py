"""world_size-2 gloo test of the multi-GPU path (CPU only; see tests/dist_worker.py)."""

import json
import os
import socket
import subprocess
import sys

from oracle.field import P

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_gloo_sharded_sumchecks(tmp_path):
    env = dict(os.environ, GKR_TEST_OUT=str(tmp_path), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(REPO, "tests", "dist_worker.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    outs = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    assert [o["units"] for o in outs] == [[0, 1, 2, 3, 4, 5], [6, 7, 8, 9, 10]]
    for o in outs:
        assert o["world"] == 2
        # (P-1) + (P-2) = 2P - 3 = P - 3 (mod P): the limb-widened sum must be reduced exactly
        assert [int(v) for v in o["allreduce"]] == [P - 3, 2 * 12345 + 1, 0]
        assert [[int(v) for v in row] for row in o["allgather"]] == [[7], [8]]
        assert o["or"] is True
        assert o["layer_ok"] and o["layer_short_ok"] and o["mle_ok"] and o["gate_sharded_ok"] and o["mle_split_ok"]
