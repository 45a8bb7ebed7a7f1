import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding as ob
    ob.lib()   # builds oracle/build/libpirip_oracle.so on first use
    return ob


@pytest.fixture(scope="session")
def built_lib():
    """libpirip_hip.so, built in-tree if missing (hipcc cross-compiles without a GPU)."""
    import pirip_amd
    if not os.path.exists(pirip_amd.lib_path()):
        pirip_amd.build()
    return pirip_amd.lib()
