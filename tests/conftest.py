"""pytest configuration: registers the ``gpu`` marker; GPU tests call through the C ABI."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import dynedge_oracle
    dynedge_oracle.build()
    return dynedge_oracle


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    out = {}
    for name in ("reference_events", "reference_known_answers", "oracle_expected"):
        path = os.path.join(GOLDEN, name + ".npz")
        if os.path.exists(path):
            out[name] = np.load(path)
    return out
