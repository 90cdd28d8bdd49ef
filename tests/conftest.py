import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFDATA = os.path.join(GOLDEN, "reference_data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """CPU and GPU suites both need the two shared objects; build them if absent."""
    import __graft_entry__ as g
    from gretel_amd import _lib
    from oracle import c_oracle
    from gretel_amd import bamio
    if not os.path.exists(_lib.SO_PATH) or not os.path.exists(c_oracle._SO) or not os.path.exists(bamio.IO_SO):
        g.build()
    yield
