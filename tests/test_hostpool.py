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


def test_worker_pool_under_thread_sanitizer():
    """`make tsan` (gkr_amd/csrc/Makefile): the same stress under -fsanitize=thread -- run_now's retire handshake and
    the session guard must be free of data races, not just of wrong results."""
    build = subprocess.run(["make", "-C", os.path.join(REPO, "gkr_amd", "csrc"), "tsan"], capture_output=True, text=True)
    if build.returncode != 0:
        pytest.skip("no ThreadSanitizer runtime for this compiler: " + build.stderr[-200:])
    exe = os.path.join(REPO, "gkr_amd", "build_san", "hostpool_stress_tsan")
    for workers in ("1", "3"):
        out = subprocess.run([exe, workers, "60000"], capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1"))
        assert out.returncode == 0 and "bad=0" in out.stdout and "ThreadSanitizer" not in out.stderr, out.stdout + out.stderr[-3000:]


def test_r1cs_compiler_parallel_stages_under_thread_sanitizer():
    """`make tsan_r1cs`: gkr_r1cs_compile's parallel tree building (sharded arena) and group compilation under
    -fsanitize=thread, 1 / 3 / 8 / 2 threads on one R1CS -- identical circuits every time, no race report
    (tests/r1cs_compile_race.cpp)."""
    build = subprocess.run(["make", "-C", os.path.join(REPO, "gkr_amd", "csrc"), "tsan_r1cs"], capture_output=True, text=True)
    if build.returncode != 0:
        pytest.skip("no ThreadSanitizer runtime for this compiler: " + build.stderr[-200:])
    exe = os.path.join(REPO, "gkr_amd", "build_san", "r1cs_compile_race_tsan")
    out = subprocess.run([exe, "3000"], capture_output=True, text=True, timeout=900, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert out.returncode == 0 and "bad=0" in out.stdout and "ThreadSanitizer" not in out.stderr, out.stdout + out.stderr[-3000:]


def test_first_use_of_the_transcript_tables_under_thread_sanitizer():
    """tests/build_host_tsan.sh builds the library's whole host side with -fsanitize=thread (libgkr_tsan.so) and
    tests/first_use_race.cpp: 32 threads make the FIRST call into the MiMC7 constant tables and the IFMA initialisation
    at the same instant, as gkr_prove_many's crew does in a fresh process.  No race report, known answers everywhere.
    (CPU only: the script does not travel to the GPU box.)"""
    script = os.path.join(HERE, "build_host_tsan.sh")
    if not os.path.exists(script):
        pytest.skip("tests/build_host_tsan.sh is not here (GPU box)")
    build = subprocess.run(["bash", script], capture_output=True, text=True)
    if build.returncode != 0:
        pytest.skip("no ThreadSanitizer build of the host side here: " + build.stderr[-300:])
    exe = os.path.join(REPO, "gkr_amd", "build_san", "first_use_race_tsan")
    for _ in range(3):
        out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
        assert out.returncode == 0 and "bad=0" in out.stdout and "ThreadSanitizer" not in out.stderr, out.stdout + out.stderr[-3000:]


def test_host_only_units_under_address_and_ub_sanitizers():
    """`make asan`: the R1CS / witness containers and compiler, the circom text generators, keccak and the IFMA hash
    driven through the C ABI with valid, truncated and bit-flipped inputs (tests/host_sanitize.cpp)."""
    build = subprocess.run(["make", "-C", os.path.join(REPO, "gkr_amd", "csrc"), "asan"], capture_output=True, text=True)
    if build.returncode != 0:
        pytest.skip("no AddressSanitizer runtime for this compiler: " + build.stderr[-200:])
    exe = os.path.join(REPO, "gkr_amd", "build_san", "host_sanitize")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert out.returncode == 0 and "host_sanitize: ok" in out.stdout, out.stdout + out.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-3000:]
