import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import __graft_entry__ as entry
    entry.ensure_built()  # the HIP library, code objects and the oracle are git-ignored build products


@pytest.fixture(scope="session")
def orc():
    import orc_binding
    return orc_binding.load()


@pytest.fixture(scope="session")
def hal():
    """A device context.  GPU tests must run the HIP library: no fallback, a missing library or device is an error."""
    import hyperfridge_r0_amd as r0
    h = r0.Hal(0)
    yield h
    h.close()


def circuit_path(name):
    return os.path.join(ROOT, "circuits", name + ".r0c")
