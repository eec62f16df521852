"""Host-side mirror of `VectorIndexManager.vectorSearch`.

Everything the reference does AROUND the seam stays on the host, in the same
order and with the same arithmetic
(/root/reference/lib/src/core/vector_index_manager.dart:475-589):

  schema/meta lookup -> empty list when nothing is indexed      (:484-508)
  query prep: _toFloat32, and _normalizeFloat32 for cosine      (:514-520, :1385-1408)
  >>> the seam: NghGraphEngine.search -> HipVectorBackend.search (:538-548)
  nodeId -> primary key, rows without a key dropped             (:553-579)
  VectorSearchResult(primaryKey, distance, score)               (:576-585, :1411-1423)
  final re-sort by distance                                     (:587)

In the product this file's role is played by the Dart class itself (see
INTEGRATION.md); here it lets the parity tests read like calls to
`db.vectorSearch(...)` (/root/reference/lib/tostore.dart:493-511).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, Optional, Sequence

import numpy as np

from .backend import HipVectorBackend, HipVectorIndex
from ._ffi import METRIC_COSINE, METRIC_IP, METRIC_L2


def to_float32(values: Sequence[float], dimensions: int) -> np.ndarray:
    """ref: core/vector_index_manager.dart:1385-1392 (query) and
    core/compute/vector_batch_prepare_compute.dart:79-86 (stored rows):
    truncate / zero-pad to `dimensions`; Float32List stores round f64 -> f32."""
    out = np.zeros(dimensions, dtype=np.float32)
    vals = np.asarray(values, dtype=np.float64).reshape(-1)
    n = min(vals.shape[0], dimensions)
    with np.errstate(over="ignore"):
        out[:n] = vals[:n].astype(np.float32)
    return out


def normalize_float32(v: np.ndarray) -> np.ndarray:
    """ref: core/vector_index_manager.dart:1395-1408.  Magnitude accumulates in
    f64 in element order; a zero vector is returned unchanged."""
    v = np.asarray(v, dtype=np.float32)
    mag = 0.0
    for x in v.tolist():  # python floats are IEEE binary64, same as Dart double
        mag += x * x
    mag = math.sqrt(mag)
    if mag == 0:
        return v
    inv = 1.0 / mag
    return (v.astype(np.float64) * inv).astype(np.float32)


def _dart_compare(a: float, b: float) -> int:
    """double.compareTo [external: Dart SDK]: NaN greatest, -0.0 < 0.0."""
    if a < b:
        return -1
    if a > b:
        return 1
    if a == b:
        if a == 0.0:
            an, bn = math.copysign(1.0, a) < 0, math.copysign(1.0, b) < 0
            return 0 if an == bn else (-1 if an else 1)
        return 0
    if math.isnan(a):
        return 0 if math.isnan(b) else 1
    return -1


def distance_to_score(distance: float, metric: int) -> float:
    """ref: core/vector_index_manager.dart:1411-1423."""
    if metric == METRIC_L2:
        return 1.0 / (1.0 + distance)
    if metric == METRIC_IP:
        try:
            return 1.0 / (1.0 + math.exp(-(-distance)))
        except OverflowError:  # Dart: exp -> inf, 1/(1+inf) = 0
            return 0.0
    s = 1.0 - distance  # .clamp(0.0, 1.0) compares with compareTo
    if _dart_compare(s, 0.0) < 0:
        return 0.0
    if _dart_compare(s, 1.0) > 0:
        return 1.0
    return s


@dataclass
class VectorSearchResult:
    """ref: model/query_result.dart:207-228."""
    primaryKey: str
    distance: float
    score: float


class VectorIndexManager:
    """One vector index (table, field) served from the GPU.

    `pk_of` plays the nodeId -> primary-key B+Tree of
    vector_index_manager.dart:556-569 (a dict or callable; missing -> row dropped).
    """

    def __init__(self, dimensions: int, metric: int, capacity_rows: int = 0, n_devices: int = 1):
        self.dimensions = dimensions
        self.metric = metric
        self.index = HipVectorIndex(dimensions, metric, capacity_rows, n_devices)
        self.backend = HipVectorBackend(self.index)
        self._pk: Dict[int, str] = {}
        self._node_of_pk: Dict[str, int] = {}
        self._next_node_id = 0

    def close(self) -> None:
        self.index.close()

    # -- write path: what reaches the vector index (inserts + deletes only;
    #    ref: core/index_manager.dart:3123-3134) ---------------------------------
    def insert_batch(self, primary_keys: Sequence[str], vectors: Sequence[Sequence[float]]) -> None:
        """ref: vector_index_manager.dart:349-419 -> ngh_graph_engine.dart:297-403:
        rows converted with _toFloat32, node ids dense from nextNodeId (:321)."""
        if len(primary_keys) != len(vectors):
            raise ValueError("primary_keys / vectors length mismatch")
        if not primary_keys:
            return
        rows = np.stack([to_float32(v, self.dimensions) for v in vectors])
        first = self._next_node_id
        self.index.append(first, rows)
        for i, pk in enumerate(primary_keys):
            self._pk[first + i] = str(pk)
            self._node_of_pk[str(pk)] = first + i
        self._next_node_id += len(primary_keys)

    def delete_batch(self, primary_keys: Sequence[str]) -> None:
        """ref: vector_index_manager.dart:421-459 -> ngh_graph_engine.dart:411-445."""
        ids = [self._node_of_pk.pop(str(pk)) for pk in primary_keys if str(pk) in self._node_of_pk]
        if ids:
            self.index.set_deleted(ids)
            for i in ids:
                self._pk.pop(i, None)

    # -- query path -----------------------------------------------------------------
    def vectorSearch(self, queryVector: Sequence[float], topK: int = 10, efSearch: Optional[int] = None,
                     distanceThreshold: Optional[float] = None, rowMask=None,
                     pk_of: Optional[Callable[[int], Optional[str]]] = None) -> list:
        """ref: core/vector_index_manager.dart:475-589."""
        if self.index.size == 0:  # :504 `meta.totalVectors == 0` -> const []
            return []
        query_f32 = to_float32(queryVector, self.dimensions)  # :514
        search_query = query_f32
        if self.metric == METRIC_COSINE:  # :516-520
            search_query = normalize_float32(query_f32)
        results = self.backend.search(query=search_query, topK=topK, efSearch=efSearch,
                                      distanceThreshold=distanceThreshold, rowMask=rowMask)  # :538
        if not results:  # :553
            return []
        sorted_by_node = sorted(results, key=lambda r: r.nodeId)  # :555-556
        lookup = pk_of if pk_of is not None else self._pk.get
        entries = []
        for r in sorted_by_node:
            pk = lookup(r.nodeId)
            if pk is None:  # :579
                continue
            entries.append(VectorSearchResult(primaryKey=pk, distance=r.distance,
                                              score=distance_to_score(r.distance, self.metric)))
        # :587 re-sort by distance; Dart's sort is not stable, ours keeps node order on ties
        import functools
        entries.sort(key=functools.cmp_to_key(lambda a, b: _dart_compare(a.distance, b.distance)))
        return entries
