import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

try:  # torch ships its own libamdhip64: it has to be the first HIP runtime the process loads, or a
    import torch  # noqa: F401  later torch.cuda init finds "no HIP GPUs" (tests that hand torch tensors to the C ABI)
except ImportError:  # pragma: no cover
    pass

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return graft.load_package()


@pytest.fixture(scope="session")
def oracle():
    o = graft.load_oracle()
    o.lib()
    return o


@pytest.fixture(scope="session")
def gpu(pkg):
    if pkg.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible (the HIP path has no fallback)")
    return 0
