import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="session")
def oracle_det():
    from oracle_py import Oracle
    return Oracle("det")


@pytest.fixture(scope="session")
def oracle_libm():
    from oracle_py import Oracle
    return Oracle("libm")


@pytest.fixture(scope="session")
def loop(pkg):
    """The HIP render loop on cuda:0 — fails loudly when the library or the GPU is missing."""
    return pkg.HipRenderLoop(0)
