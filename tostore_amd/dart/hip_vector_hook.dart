// hip_vector_hook.dart -- the hook a ToStore maintainer applies to VectorIndexManager
// (lib/src/core/vector_index_manager.dart) so that vectorSearch() is answered by libtostore_hip.so whenever a
// device copy of the index exists, and by the original NghGraphEngine.search otherwise.
//
// Add this file as lib/src/core/hip_vector_hook.dart next to the bridge (lib/src/handler/hip_vector_backend.dart =
// tostore_hip_bridge.dart) and change VectorIndexManager in FOUR places -- each is one line that calls into the
// mixin, listed at the bottom of this file with the reference's own line numbers:
//
//   class VectorIndexManager with HipVectorHook { ... }
//
// NOT compiled in this repository (the build image has no Dart SDK).  tests/test_dart_bridge.py checks what can be
// checked without one: every HipVectorBackend member this file calls exists in the bridge with the arity used
// here, and the four call sites below name reference lines that hold what they say.  The Python mirror of the same
// flow -- tostore_amd/vector_index_manager.py -- is what the test-suite drives end to end.
//
// Contract kept from the reference (SURVEY.md section 8b):
//   * everything above the seam (schema / meta lookup, _toFloat32, _normalizeFloat32: :483-520) and below it
//     (nodeId -> primary key, _distanceToScore, final sort: :553-588) runs unchanged;
//   * any native failure (library missing, no GPU, TSH_E_*) returns null and the caller falls through to the Dart
//     graph search -- the catch-and-fall-back style of lib/src/handler/system_ffi_helper.dart:21-55;
//   * the device copy follows the index: appended to after insertBatch, tombstoned after deleteBatch, dropped
//     whenever the reference drops its own caches for the index and after reorderByLocality (node ids are
//     renumbered there: a stale copy would return wrong rows).

import 'dart:typed_data';

import '../handler/hip_vector_backend.dart';
import '../handler/logger.dart';
import '../model/ngh_index_meta.dart';
import 'ngh_graph_engine.dart' show NghSearchResult;

/// What the hook needs from its host class (VectorIndexManager already has both: `_dataStore.pathManager`
/// (core/data_store_impl.dart:180) and `_dataStore.maxEntriesPerDir` (:189-190)).
abstract class HipVectorHookHost {
  /// `<index>/ngh` of (tableName, indexName): PathManager.getNghIndexPath (core/path_manager.dart:275-278).
  Future<String> hipNghIndexPath(String tableName, String indexName);

  /// DataStoreImpl.maxEntriesPerDir (core/data_store_impl.dart:189-190; default 500, handler/common.dart:43).
  int get hipMaxEntriesPerDir;
}

mixin HipVectorHook implements HipVectorHookHost {
  /// Device copies, key = '$tableName/$indexName' (the key format of _metaLoadingFutures, :35-37).
  final Map<String, HipVectorBackend> _hip = {};

  /// Indexes whose device copy could not be made (no library, no GPU, holes on disk): not retried on every query.
  final Set<String> _hipRefused = {};

  /// In-flight cold loads: concurrent first searches share one (the coalescing of _loadMeta, :596-620).
  final Map<String, Future<HipVectorBackend?>> _hipLoading = {};

  /// Searches above this many stored floats use the ticket form (tsh_search_submit / _ready / _wait) so that the
  /// isolate goes back to its event loop while the GPU scans: 256 M floats ~ 1 GB ~ 0.15 ms at 7 TB/s, far inside
  /// the 8 ms client budget (model/data_store_config.dart:225-230); a 10 M x 1536 shard is 8.8 ms and must not block.
  static const int _hipAsyncAboveFloats = 1 << 28;

  /// The reference caps what a search returns: `ef = min(efSearch ?? meta.efSearch, max(topK * 5, 32))`
  /// (ngh_graph_engine.dart:80-82) is also the capacity of its result heap (:168), so k = 100 with the default
  /// efSearch = 64 (model/ngh_index_meta.dart:196) yields at most 64 rows, and an index without a medoid yields none
  /// (:78).  The device path has no ef -- it returns the k best of ALL rows.  A maintainer who wants the old row
  /// COUNT back (a caller that pages on "fewer than topK rows = end of data", a test that pins the length) sets this
  /// for the whole manager, or names single indexes in [hipHonourEfCapFor] ('table/index'): the device answer is then
  /// cut to `min(topK, ef)` rows -- still the exact nearest ones, which the reference's own 64 need not be.
  bool hipHonourEfCap = false;
  final Set<String> hipHonourEfCapFor = {};

  String _hipKey(String tableName, String indexName) => '$tableName/$indexName';

  /// The device copy of an index, made on first use: ONE native call reads meta.json, every rawvec partition and
  /// the graph slots' deleted flags (tsh_index_open_ngh; the bridge refuses a copy with absent pages).
  /// null = use the Dart path.
  Future<HipVectorBackend?> hipFor(String tableName, String indexName, NghIndexMeta meta) {
    final key = _hipKey(tableName, indexName);
    final cur = _hip[key];
    if (cur != null) return Future.value(cur);
    if (_hipRefused.contains(key) || !HipVectorBackend.available) return Future.value(null);
    final loading = _hipLoading[key];
    if (loading != null) return loading;
    final f = () async {
      try {
        final nghDir = await hipNghIndexPath(tableName, indexName);
        final b = HipVectorBackend.tryOpen(nghDir, meta, hipMaxEntriesPerDir);
        if (b == null) {
          _hipRefused.add(key);
          return null;
        }
        // the copy must describe the index as the caller sees it NOW: rows flushed after the files were read arrive
        // through hipAfterInsert; a copy that is AHEAD of meta (cannot happen: meta is persisted last, :401) is refused
        if (b.size > meta.nextNodeId) {
          Logger.warn('device copy of $key holds ${b.size} rows, meta says ${meta.nextNodeId}: not used',
              label: 'HipVectorHook');
          b.dispose();
          _hipRefused.add(key);
          return null;
        }
        _hip[key] = b;
        return b;
      } catch (e) {
        Logger.warn('device copy of $key failed: $e', label: 'HipVectorHook');
        _hipRefused.add(key);
        return null;
      } finally {
        _hipLoading.remove(key);
      }
    }();
    _hipLoading[key] = f;
    return f;
  }

  /// CALL SITE 1 -- vectorSearch(), around `_graphEngine.search(...)` (:536-551).  Returns null when the Dart graph
  /// search must answer (no device copy, or the native call failed).  `searchQuery` is the Float32List after
  /// _toFloat32 and, for cosine, _normalizeFloat32 (:514-520): the library does not normalise.
  Future<List<NghSearchResult>?> hipSearch(String tableName, String indexName, NghIndexMeta meta,
      Float32List searchQuery, int topK, double? distanceThreshold, {int? efSearch}) async {
    final capped = hipHonourEfCap || hipHonourEfCapFor.contains(_hipKey(tableName, indexName));
    // (ngh_graph_engine.dart:78: the reference answers nothing while the graph has no entry point)
    if (capped && meta.medoidNodeId < 0) return const [];
    if (capped) {
      final efRaw = efSearch ?? meta.efSearch; // :80
      final ef = efRaw < (topK * 5 > 32 ? topK * 5 : 32) ? efRaw : (topK * 5 > 32 ? topK * 5 : 32); // :82
      if (ef < topK) topK = ef < 0 ? 0 : ef; // the reference's result heap holds ef entries (:168)
      if (topK <= 0) return const [];
    }
    final hip = await hipFor(tableName, indexName, meta);
    if (hip == null) return null;
    // a copy that lags the index (an append failed, see hipAfterInsert) is never searched
    if (hip.size != meta.nextNodeId) {
      hipDrop(tableName, indexName);
      return null;
    }
    if (meta.nextNodeId * meta.dimensions > _hipAsyncAboveFloats) {
      return hip.searchAsync(searchQuery, topK, distanceThreshold: distanceThreshold);
    }
    return hip.search(searchQuery, topK, distanceThreshold: distanceThreshold);
  }

  /// CALL SITE 2a -- writeChanges(), after insertBatch + _partitionManager.writeChanges succeeded (:368-388):
  /// `vectors` are the Float32Lists of _prepareInsertVectorsBatch (:349-356), node ids dense from `startNodeId`
  /// (:359, ngh_graph_engine.dart:321).  A failed append drops the copy (it would lag the index).
  void hipAfterInsert(String tableName, String indexName, int startNodeId, List<Float32List> vectors) {
    final key = _hipKey(tableName, indexName);
    _hipRefused.remove(key); // an index that had nothing to load may have now
    final hip = _hip[key];
    if (hip == null || vectors.isEmpty) return;
    if (hip.size != startNodeId || !hip.append(startNodeId, vectors)) {
      hipDrop(tableName, indexName);
    }
  }

  /// CALL SITE 2b -- writeChanges(), after deleteBatch succeeded (:429-434): the same node ids.  Deleted rows are
  /// never returned by the device path (the reference's beam search can still leak them from other pages,
  /// ngh_graph_engine.dart:230-232: documented divergence).
  void hipAfterDelete(String tableName, String indexName, List<int> nodeIds) {
    final hip = _hip[_hipKey(tableName, indexName)];
    if (hip == null || nodeIds.isEmpty) return;
    if (!hip.setDeleted(nodeIds)) hipDrop(tableName, indexName);
  }

  /// CALL SITE 3 -- reorderByLocality(), where it returns true (:1154-1158: node ids were renumbered), and
  /// clearCacheForIndex (:1198-1202).
  void hipDrop(String tableName, String indexName) {
    final key = _hipKey(tableName, indexName);
    _hipRefused.remove(key);
    _hip.remove(key)?.dispose();
  }

  /// CALL SITE 4a -- clearCacheForTable (:1192-1195).
  void hipDropTable(String tableName) {
    final prefix = '$tableName/';
    for (final key in _hip.keys.where((k) => k.startsWith(prefix)).toList()) {
      _hip.remove(key)?.dispose();
    }
    _hipRefused.removeWhere((k) => k.startsWith(prefix));
  }

  /// CALL SITE 4b -- dispose() (:1205-1216).
  void hipDisposeAll() {
    for (final b in _hip.values) {
      b.dispose();
    }
    _hip.clear();
    _hipRefused.clear();
  }

  /// Logger diagnostics (handler/logger.dart:8-60): the library's counters per device copy.
  Map<String, Map<String, num>?> hipCounters() => {for (final e in _hip.entries) e.key: e.value.counters()};
}

// ---------------------------------------------------------------------------------------------------------------
// The four edits to lib/src/core/vector_index_manager.dart (v3.2.0), as a maintainer would make them:
//
//   class VectorIndexManager with HipVectorHook {                                     // :28
//     @override
//     Future<String> hipNghIndexPath(String t, String i) => _dataStore.pathManager.getNghIndexPath(t, i);
//     @override
//     int get hipMaxEntriesPerDir => _dataStore.maxEntriesPerDir;
//
//   1. vectorSearch(), replacing `List<NghSearchResult> results; try { results = await _graphEngine.search(` (:536-551):
//        List<NghSearchResult>? hipResults =
//            await hipSearch(tableName, indexName, meta, searchQuery, topK, distanceThreshold, efSearch: efSearch);
//        List<NghSearchResult> results;
//        if (hipResults != null) { lease?.release(); results = hipResults; } else { /* the original try / finally */ }
//
//   2. writeChanges(): after `meta = await _partitionManager.writeChanges(... dirtyRawVectorPages ...)` (:378-388):
//        hipAfterInsert(tableName, indexName, startNodeId, vectors);
//      and after `final result = await _graphEngine.deleteBatch(...)` (:429-434):
//        hipAfterDelete(tableName, indexName, nodeIdsToDelete);
//
//   3. reorderByLocality(), before `return true;` (:1158):          hipDrop(tableName, indexName);
//      clearCacheForIndex (:1198):                                    hipDrop(tableName, indexName);
//
//   4. clearCacheForTable (:1192):                                    hipDropTable(tableName);
//      dispose() (:1205):                                             hipDisposeAll();
//
// Row UPDATES need nothing: IndexManager.writeChanges hands VectorIndexManager.writeChanges inserts and deletes only
// (core/index_manager.dart:3123-3134).
// ---------------------------------------------------------------------------------------------------------------
