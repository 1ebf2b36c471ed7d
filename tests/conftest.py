import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A process that uses both torch's GPU runtime (the device exchange tests: CUDA tensors, RCCL) and the library
    must bring torch's up FIRST: torch ships its own HIP runtime, and on this image it finds no device once the
    system's runtime (which libgkr_amd.so links) has initialised in the process.  bench.py has the same order
    (process group and torch first, then the first gkr context)."""
    if any(it.get_closest_marker("gpu") for it in items):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def gkr_cases():
    return load_golden("gkr_circuits.json")["cases"]


@pytest.fixture(scope="session")
def mle_cases():
    return load_golden("mle_sumcheck.json")["cases"]
