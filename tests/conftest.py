import os
import sys

import pytest

try:  # PyTorch bundles its own HIP runtime: load it before libminlz_hip.so pulls in /opt/rocm's, whatever the test order
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return GOLDEN


@pytest.fixture(scope="session")
def twain():
    return open(os.path.join(GOLDEN, "Mark.Twain-Tom.Sawyer.txt"), "rb").read()


@pytest.fixture(scope="session")
def twain_mzb():
    return open(os.path.join(GOLDEN, "Mark.Twain-Tom.Sawyer.txt.mzb"), "rb").read()


@pytest.fixture(scope="session")
def ctx():
    """HIP context; GPU tests fail loudly (no CPU fallback) when the extension or device is missing."""
    import minlz_amd as mz
    c = mz.Context(0)
    yield c
    c.close()
