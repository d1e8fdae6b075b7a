import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "optimal-control-dynamic-programming_amd"
for p in (str(PKG), str(ROOT)):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure libhjbdp.so and the oracle's C twin exist (cross-compiles here)."""
    import __graft_entry__ as g
    g.build()
    return g


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(ROOT / "tests" / "golden" / "obj_1.npz")
