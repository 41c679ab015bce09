import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu")


@pytest.fixture(scope="session")
def built():
    """Build the HIP engine and the C oracle once per session (hipcc cross-compiles on CPU)."""
    import __graft_entry__ as ge
    ge.build()
    return True


@pytest.fixture(scope="session")
def ctx(built):
    import dvs_mcemvs_amd as d
    c = d.Context(0)  # raises DsiError(NO_DEVICE) without a gfx950 GPU: no fallback
    yield c
    c.close()
