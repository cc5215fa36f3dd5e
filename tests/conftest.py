import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Both shared objects exist (the driver runs __graft_entry__.build() first; this keeps a bare pytest working)."""
    import magnetite_amd
    import oracle
    if not os.path.exists(magnetite_amd._lib.SO_PATH):
        magnetite_amd.build()
    oracle.build()
    return True
