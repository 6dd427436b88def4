import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """Make sure libvacnic_hip.so exists (cross-compiles without a GPU)."""
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "vacnic_amd", "libvacnic_hip.so")):
        g.build()
    return os.path.join(ROOT, "vacnic_amd", "libvacnic_hip.so")
