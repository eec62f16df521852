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


@pytest.fixture(params=["exact", "prefilter"])
def scan_path(request, monkeypatch):
    """Searches that look at no more than 16384 rows answer from the exact sums of all of them (tsh_exact.hip.h);
    everything else goes through the f32 pre-filter (scan, select, re-rank).  Tests on small shapes run both ways:
    "prefilter" switches the exact path off for every index the test creates (TSH_OPT_EXACT_SCAN_ROWS = 0), so the
    pre-filter kernels keep their coverage of small inputs, ties and edges.  A module asks for it with
    pytestmark = [..., pytest.mark.usefixtures("scan_path")]."""
    if request.param == "prefilter":
        from tostore_amd import backend

        init = backend.HipVectorIndex.__init__

        def patched(self, *a, **kw):
            init(self, *a, **kw)
            self.set_exact_scan_rows(0)

        monkeypatch.setattr(backend.HipVectorIndex, "__init__", patched)
    return request.param
