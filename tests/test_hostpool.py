"""SpinPool (gkr_amd/csrc/hostpool.h), the host transcript's worker pool: a stress of run_now with jobs whose state
lives in the caller's stack frame (tests/hostpool_stress.cpp).  A worker calling a retired job would show up as a
poisoned frame.  (The retire-then-check handshake needs sequentially consistent accesses on both sides; with
weaker ordering it crashed on the GPU box with one worker and short layer rounds.)"""

import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def stress_binary(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("hostpool") / "hostpool_stress")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(REPO, "gkr_amd", "csrc"),
                           os.path.join(HERE, "hostpool_stress.cpp"), "-o", exe])
    return exe


@pytest.mark.parametrize("workers", [1, 3])
def test_run_now_never_calls_a_retired_job(stress_binary, workers):
    out = subprocess.run([stress_binary, str(workers), "2000000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "bad=0" in out.stdout, out.stdout + out.stderr
