"""pytest configuration: markers, paths, and the golden-fixture loader."""
import json
import os
import sys

import pytest

# This image carries two ROCm runtimes with the same sonames (/opt/rocm and the copy bundled in torch/lib).
# Whichever libamdhip64 is loaded first serves the whole process; torch only finds the device through its
# own copy, so tests that use torch next to libmc_mi355x.so need torch loaded first.
try:
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def fromhex(h):
    return float.fromhex(h)


@pytest.fixture(scope="session")
def po():
    """The CPU oracle bindings (test infrastructure)."""
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle
