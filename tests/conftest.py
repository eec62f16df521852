import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; GPU tests fail loudly when it is missing or sees no device."""
    # torch bundles its own ROCm runtime: when both live in one process torch must
    # initialise first, or its HSA copy finds the device already claimed
    try:
        import torch

        torch.cuda.is_available() and torch.cuda.init()
    except ImportError:
        pass
    from tostore_amd import _ffi

    L = _ffi.lib()
    assert L.tsh_device_count() >= 1, "libtostore_hip.so sees no HIP device"
    return L
