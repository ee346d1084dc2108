import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def gsref():
    """The CPU oracle (test infrastructure)."""
    from oracle import gsref as module

    module.build()
    return module


@pytest.fixture(scope="session")
def gs():
    """The product package; its HIP library must exist (no fallback)."""
    import cudagaussianrenderer_amd as module

    module.lib()
    return module
