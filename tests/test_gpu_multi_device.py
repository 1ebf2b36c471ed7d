"""What has to hold with MORE THAN ONE device ("everything N > 1 has only run degenerate"): the
library-owned RCCL exchange with two real ranks, one host process dealing gkr_prove_many over two devices, and what every
host can check on one GPU as well -- the calling thread's current device is the caller's, and a rank whose peer never
arrives waits inside gkr_exchange_rccl_create (documented in include/gkr_amd.h) instead of crashing or returning garbage.
The two-device cases skip where fewer than two GPUs are visible (every builder's box so far)."""
import ctypes
import os
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _device_count():
    import torch
    return torch.cuda.device_count()      # (does not initialise the GPU)


def _hip():
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6", "/opt/rocm/lib/libamdhip64.so"):
        try:
            return ctypes.CDLL(name)
        except OSError:
            pass
    pytest.skip("libamdhip64 not loadable through ctypes")


def _current_device(hip):
    d = ctypes.c_int(-1)
    assert hip.hipGetDevice(ctypes.byref(d)) == 0
    return d.value


def test_rccl_create_with_an_absent_rank_waits_and_can_be_ended():
    """Rank 0 of a world of two, alone: gkr_exchange_rccl_create does not return (ncclCommInitRank has no timeout) -- the
    child is still inside the call after 20 s, has not crashed, and ends when its parent terminates it.  The parent kills
    exactly the process it started; nothing is re-executed."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    child = subprocess.Popen([sys.executable, os.path.join(HERE, "multi_device_worker.py"), "rccl-alone"], env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True)
    try:
        line = child.stdout.readline()
        assert "CREATING" in line, line + child.stderr.read()
        t0 = time.time()
        while time.time() - t0 < 20 and child.poll() is None:
            time.sleep(0.5)
        assert child.poll() is None, "the call returned or the process died: rc %s\n%s" % (child.returncode, child.stderr.read()[-2000:])
    finally:
        child.kill()
        child.wait(timeout=60)


def test_the_callers_current_device_survives_every_call():
    """gkr_ctx_create, gkr_ctx_create_multi and gkr_prove_many (child contexts on other devices, member 0 without an item)
    leave the calling thread's current device as they found it."""
    from gkr_amd import Context, GKRCircuit, Layer
    hip = _hip()
    n = _device_count()
    home = n - 1                      # the last device: not the one the contexts below live on when there are several
    assert hip.hipSetDevice(ctypes.c_int(home)) == 0
    rng = np.random.default_rng(5)
    circuit = GKRCircuit([Layer(3, rng.integers(0, 2, 8).tolist(), rng.integers(0, 16, 8).tolist(), rng.integers(0, 16, 8).tolist())], 4)
    wit = np.zeros((1, 16, 4), dtype=np.uint64)
    wit[0, :, 0] = np.arange(1, 17)
    devs = [0, 0]        # (child contexts on device 0 while the caller's current device is the LAST one: a real second device is
                         # the business of the tests below, which have never met one)
    with Context(0) as one:
        assert _current_device(hip) == home
        with Context(devices=devs) as multi:
            assert _current_device(hip) == home
            work = [(circuit, wit)]
            want = one.prove_many_raw(one.prepare_many(work), 1)
            assert _current_device(hip) == home
            # max_concurrent 4 > one item: member 0 may get no item, the children are created on the other devices
            got = multi.prove_many_raw(multi.prepare_many(work), 4)
            assert _current_device(hip) == home
            for a, b in zip(got[0], want[0]):
                assert np.array_equal(a, b)


# The two tests below need a second GPU (no builder's box had one): they skip where there is none and FAIL like any other test
# where there is one -- a broken RCCL or multi-device path must turn the first multi-GPU run red.
@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs")
def test_native_rccl_exchange_with_two_ranks_on_two_devices(tmp_path):
    """gkr_exchange_rccl_create with nranks = 2, a process per GPU, the id handed over in a file: the gate-sharded layer
    sumcheck (two all-reduces per sumcheck) and the split plain sumcheck (one per pass + the gather) over xGMI, every
    rank's transcript equal to the C checker's."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    path = str(tmp_path / "rccl_id")
    kids = [subprocess.Popen([sys.executable, os.path.join(HERE, "multi_device_worker.py"), "rccl-rank", path, str(r), "2", str(r)], env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    try:
        outs = [k.communicate(timeout=600) for k in kids]
    finally:
        for k in kids:
            if k.poll() is None:
                k.kill()
    for r, (k, (so, se)) in enumerate(zip(kids, outs)):
        assert k.returncode == 0 and ("OK rank %d" % r) in so, so[-2000:] + se[-3000:]


@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs")
def test_prove_many_over_two_devices_matches_one_device():
    """gkr_ctx_create_multi([0, 1]): the items of one gkr_prove_many call dealt over child contexts on BOTH devices (pinned
    records, circuit caches and streams per device) give the proofs one device gives."""
    from gkr_amd import Context, synth
    from gkr_amd.aggregate import ProvingStep
    from gkr_amd.field import as_limbs
    step = ProvingStep(synth.mimc7_demo_r1cs(nrounds=8))
    wits = np.stack([as_limbs(synth.mimc7_demo_witness(a, b, nrounds=8)) for a, b in ((2, 3), (3, 4), (5, 6))])
    inputs = step.inputs_for(wits)
    work = list(zip(step.circuits, inputs))
    with Context(0) as one, Context(devices=[0, 1]) as two:
        want = one.prove_many_raw(one.prepare_many(work), 0)
        want = [[a.copy() for a in arrs] for arrs in want]
        for threads in (0, 2, 7):
            got = two.prove_many_raw(two.prepare_many(work), threads)
            for j in range(len(work)):
                for a, b in zip(got[j], want[j]):
                    assert np.array_equal(a, b), (threads, j)
    step.close()
