import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` through gpurun)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """The CPU suite needs the oracle and (for symbol / plan tests) libxmpi.so; build in-tree."""
    from mpi_amd import build
    build.build_all()
    yield
