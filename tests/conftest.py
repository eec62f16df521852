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


@pytest.fixture(params=["exact", "exact_select", "prefilter"])
def scan_path(request, monkeypatch):
    """Searches that look at no more than 16384 rows answer from the exact sums of all of them (tsh_exact.hip.h);
    everything else goes through the f32 pre-filter (scan, select, re-rank).  Tests on small shapes run both ways:
    "prefilter" switches the exact path off for every index the test creates (TSH_OPT_EXACT_SCAN_ROWS = 0), so the
    pre-filter kernels keep their coverage of small inputs, ties and edges; "exact_select" keeps the exact scan but
    follows it with the one-workgroup select (TSH_OPT_EXACT_SELECT = 0) instead of the wide pick, which is also what a
    pick falls back to when its cut bin overflows.  A module asks for it with
    pytestmark = [..., pytest.mark.usefixtures("scan_path")]."""
    if request.param != "exact":
        from tostore_amd import backend

        init = backend.HipVectorIndex.__init__
        param = request.param

        def patched(self, *a, **kw):
            init(self, *a, **kw)
            if param == "prefilter":
                self.set_exact_scan_rows(0)
            else:  # the exact scan followed by round 5's one-workgroup select instead of the wide pick
                self.set_exact_select(False)

        monkeypatch.setattr(backend.HipVectorIndex, "__init__", patched)
    return request.param


@pytest.fixture(params=["pointer", "handle"])
def mask_form(request, monkeypatch):
    """A row mask reaches the library as a POINTER (tsh_search's row_mask: sliced, counted and listed on the host, call
    by call) or as a HANDLE (tsh_mask_create: uploaded once, listed on the device, resident).  Tests that pass masks run
    both ways: "handle" makes HipVectorIndex.search / submit turn every mask array into a mask handle first -- one per
    distinct bitmap and index, kept until the index is closed, so handles are also reused across calls, appends and
    deletes exactly as the tests' pointer masks are.  Results must be identical.  A test asks for it with
    @pytest.mark.usefixtures("mask_form")."""
    if request.param == "handle":
        import numpy as np

        from tostore_amd import backend

        cls = backend.HipVectorIndex
        search, submit, close = cls.search, cls.submit, cls.close

        def handle_for(self, row_mask):
            if row_mask is None or isinstance(row_mask, backend.HipMask):
                return row_mask
            arr = np.ascontiguousarray(row_mask, dtype=np.uint8).reshape(-1)
            if arr.shape[0] < (self.size + 7) // 8:
                return row_mask  # (too short for the pointer form: let mask_arg say so)
            cache = self.__dict__.setdefault("_test_masks", {})
            key = arr.tobytes()
            if key not in cache:
                if len(cache) >= 6:
                    cache.pop(next(iter(cache))).close()
                cache[key] = backend.HipMask(self, arr)
            return cache[key]

        def p_search(self, queries, k, distance_threshold=None, row_mask=None):
            return search(self, queries, k, distance_threshold, handle_for(self, row_mask))

        def p_submit(self, query, k, row_mask=None):
            return submit(self, query, k, handle_for(self, row_mask))

        def p_close(self):
            for m in self.__dict__.pop("_test_masks", {}).values():
                m.close()
            close(self)

        monkeypatch.setattr(cls, "search", p_search)
        monkeypatch.setattr(cls, "submit", p_submit)
        monkeypatch.setattr(cls, "close", p_close)
    return request.param
