import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand with gcc."""
    from oracle import rtx_oracle
    rtx_oracle.build()
    rtx_oracle.lib()
    return rtx_oracle


@pytest.fixture(scope="session")
def rtx():
    """The product package; the HIP library must already be built in-tree (no JIT, no fallback)."""
    import rust_raytracing_amd
    rust_raytracing_amd.load_library()
    return rust_raytracing_amd
