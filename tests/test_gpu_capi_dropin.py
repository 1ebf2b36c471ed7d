"""The C ABI driven from a plain-C host in the reference's own argument types (tests/capi_dropin.c: wire vectors + term lists ->
gkr_layer_from_wires / gkr_values_from_terms -> gkr_prove -> gkr_verify), compared with the REFERENCE's Python prover on the toy
circuit of its own test (tests/golden/gkr_circuits.json[test_gkr_toy_z0_zero], made by tests/golden/make_golden.py)."""
import json
import os
import subprocess

import pytest

from helpers import ints, right_aligned_equal, terms_as_set

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "capi_dropin")
    lib = os.path.join(REPO, "gkr_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"), os.path.join(REPO, "tests", "capi_dropin.c"),
                           "-L", lib, "-lgkr_amd", "-Wl,-rpath," + lib, "-o", exe])
    return exe


def test_the_c_host_builds_against_the_header_alone(tmp_path):
    """(CPU) the program compiles as C99 against include/gkr_amd.h and links the library: every entry point it uses exists with
    the declared signature.  Without a GPU it stops at gkr_ctx_create with the library's own message."""
    exe = _build(tmp_path)
    import torch
    if torch.cuda.device_count() == 0:
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert out.returncode == 1 and "gkr_ctx_create" in out.stderr, (out.returncode, out.stderr)


@pytest.mark.gpu
def test_c_host_in_reference_types_matches_the_reference_python_prover(tmp_path, gkr_cases):
    case = next(c for c in gkr_cases if c["name"] == "test_gkr_toy_z0_zero")
    out = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads(out.stdout)
    assert got["verifier_accepts"] is True and got["k"] == case["k"] and got["depth"] == len(case["k"])
    # the adapters decoded the reference's wire vectors and term list into the fixture's gate arrays and input values
    assert got["gates"] == case["layers"] and ints(got["input_values"]) == ints(case["inputs"])
    assert ints(got["sumcheck_r"]) == ints(case["sumcheck_r"]) and ints(got["z"]) == ints(case["z"]) and ints(got["r"]) == ints(case["r"])
    for lay in range(len(case["q"])):
        assert right_aligned_equal(ints(got["q"][lay]), ints(case["q"][lay]))
        assert all(right_aligned_equal(a, b) for a, b in zip(ints(got["sumcheck_proofs"][lay]), ints(case["sumcheck_proofs"][lay])))
    assert terms_as_set(ints(got["d"])) == terms_as_set(ints(case["D"]))
    assert terms_as_set(ints(got["input_func"])) == terms_as_set(ints(case["input_func"]))
