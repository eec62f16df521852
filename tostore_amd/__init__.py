"""tostore_amd -- MI355X-native exhaustive kNN behind ToStore's vectorSearch().

Only what the hot path needs: the HIP kernels + C-ABI (csrc/, built into
libtostore_hip.so), the ctypes binding (_ffi), and the host-side mirror of the
reference interface around the seam (backend, vector_index_manager, sharded).
"""
from . import _ffi  # noqa: F401
from .backend import HipMask, HipVectorBackend, HipVectorIndex, NghSearchResult  # noqa: F401
from .vector_index_manager import (VectorIndexManager, VectorSearchResult,  # noqa: F401
                                   distance_to_score, normalize_float32, to_float32)

__all__ = [
    "HipMask", "HipVectorBackend", "HipVectorIndex", "NghSearchResult", "VectorIndexManager",
    "VectorSearchResult", "distance_to_score", "normalize_float32", "to_float32",
]
