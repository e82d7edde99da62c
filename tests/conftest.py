import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle module (built on demand; it is test infrastructure)."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def rt():
    """The product package; its HIP library must already be built in-tree."""
    import raytracer_amd
    raytracer_amd.lib()
    return raytracer_amd
