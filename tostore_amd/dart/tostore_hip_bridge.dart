// tostore_hip_bridge.dart -- the `dart:ffi` binding a ToStore maintainer adds as
// lib/src/handler/hip_vector_backend.dart to route vectorSearch() through
// libtostore_hip.so (include/tostore_hip.h).
//
// NOT compiled in this repository: the build image has no Dart SDK.  What CAN be
// checked without one is: tests/test_dart_bridge.py parses every `typedef ...C`
// below, the symbol each one is looked up as, and the TshNghInfo struct, and
// compares arity, argument widths and field layout with include/tostore_hip.h.
// It is a mechanical mapping of the C-ABI, written in the style of the
// reference's only existing FFI user, lib/src/handler/system_ffi_helper.dart
// (DynamicLibrary.open + lookupFunction, int32 status with 0 = success,
// calloc/free in try/finally, every failure falls back to the Dart path).
// The ctypes binding tostore_amd/_ffi.py exercises the same entry points with
// the same argument meaning and IS tested (tests/test_abi.py, tests/test_gpu_*).

import 'dart:ffi';
import 'dart:math' show Random;
import 'dart:typed_data';

import 'package:ffi/ffi.dart';

import '../core/ngh_graph_engine.dart' show NghSearchResult;
import '../core/vector_quantizer.dart' show PqCodebook;
import '../model/ngh_index_meta.dart';
import '../model/table_schema.dart' show VectorDistanceMetric;
import 'logger.dart';

// ---- native signatures (include/tostore_hip.h) ---------------------------------
typedef _AbiVersionC = Int32 Function();
typedef _AbiVersionD = int Function();
typedef _DeviceCountC = Int32 Function();
typedef _DeviceCountD = int Function();
typedef _LastErrorC = Int32 Function(Pointer<Utf8>, Int32);
typedef _LastErrorD = int Function(Pointer<Utf8>, int);
typedef _CreateC = Int32 Function(Int32, Int32, Int64, Int32, Pointer<Pointer<Void>>);
typedef _CreateD = int Function(int, int, int, int, Pointer<Pointer<Void>>);
typedef _DestroyC = Int32 Function(Pointer<Void>);
typedef _DestroyD = int Function(Pointer<Void>);
typedef _AppendC = Int32 Function(Pointer<Void>, Int64, Int64, Pointer<Float>);
typedef _AppendD = int Function(Pointer<Void>, int, int, Pointer<Float>);
typedef _SetDeletedC = Int32 Function(Pointer<Void>, Pointer<Int64>, Int64);
typedef _SetDeletedD = int Function(Pointer<Void>, Pointer<Int64>, int);
typedef _LoadRawvecC = Int32 Function(
    Pointer<Void>, Pointer<Utf8>, Int32, Int32, Int64, Int64, Pointer<Int64>);
typedef _LoadRawvecD = int Function(
    Pointer<Void>, Pointer<Utf8>, int, int, int, int, Pointer<Int64>);
typedef _SizeC = Int64 Function(Pointer<Void>);
typedef _SizeD = int Function(Pointer<Void>);
typedef _SearchC = Int32 Function(Pointer<Void>, Pointer<Float>, Int32, Int32, Double,
    Pointer<Uint8>, Pointer<Int64>, Pointer<Double>, Pointer<Int32>);
typedef _SearchD = int Function(Pointer<Void>, Pointer<Float>, int, int, double,
    Pointer<Uint8>, Pointer<Int64>, Pointer<Double>, Pointer<Int32>);
typedef _OpenNghC = Int32 Function(
    Pointer<Utf8>, Int32, Int32, Pointer<Pointer<Void>>, Pointer<TshNghInfo>);
typedef _OpenNghD = int Function(
    Pointer<Utf8>, int, int, Pointer<Pointer<Void>>, Pointer<TshNghInfo>);
typedef _PqEncodeC = Int32 Function(
    Pointer<Void>, Int64, Int64, Pointer<Float>, Int32, Int32, Pointer<Uint8>);
typedef _PqEncodeD = int Function(
    Pointer<Void>, int, int, Pointer<Float>, int, int, Pointer<Uint8>);
typedef _PqTrainC = Int32 Function(Int32, Pointer<Float>, Int64, Int32, Int32, Int32, Int32,
    Pointer<Int32>, Pointer<Float>);
typedef _PqTrainD = int Function(int, Pointer<Float>, int, int, int, int, int,
    Pointer<Int32>, Pointer<Float>);

/// `tsh_ngh_info` (include/tostore_hip.h): what tsh_index_open_ngh found.  Field order and
/// widths are checked against the header by tests/test_dart_bridge.py.
final class TshNghInfo extends Struct {
  @Int32()
  external int dimensions;
  @Int32()
  external int metric;
  @Int32()
  external int precision;
  @Int32()
  external int pageSize;
  @Int32()
  external int maxDegree;
  @Int32()
  external int reserved;
  @Int64()
  external int nextNodeId;
  @Int64()
  external int totalVectors;
  @Int64()
  external int deletedCount;
  @Int64()
  external int maxPartitionFileSize;
  @Int64()
  external int rowsLoaded;
  @Int64()
  external int tombstones;
  @Int64()
  external int filesRead;
  @Int64()
  external int pagesAbsent;
  @Int64()
  external int filesAbsent;
}

/// One device-resident copy of an NGH index's raw-vector column.
///
/// Lifetime: created lazily on the first vectorSearch / writeChanges of
/// (tableName, indexName); dropped from clearCacheForTable / clearCacheForIndex /
/// dispose (vector_index_manager.dart:1192-1216) and after reorderByLocality
/// (:932-1159), which renumbers node ids.
final class HipVectorBackend {
  static DynamicLibrary? _lib;
  static bool _probed = false;

  static late final _AbiVersionD _abiVersion;
  static late final _DeviceCountD _deviceCount;
  static late final _LastErrorD _lastError;
  static late final _CreateD _create;
  static late final _DestroyD _destroy;
  static late final _AppendD _append;
  static late final _SetDeletedD _setDeleted;
  static late final _LoadRawvecD _loadRawvec;
  static late final _SizeD _size;
  static late final _SearchD _search;
  static late final _OpenNghD _openNgh;
  static late final _PqEncodeD _pqEncode;
  static late final _PqTrainD _pqTrain;

  /// True when libtostore_hip.so is loadable, ABI-compatible and sees a GPU.
  static bool get available {
    if (_probed) return _lib != null;
    _probed = true;
    try {
      final lib = DynamicLibrary.open('libtostore_hip.so');
      _abiVersion = lib.lookupFunction<_AbiVersionC, _AbiVersionD>('tsh_abi_version');
      _deviceCount = lib.lookupFunction<_DeviceCountC, _DeviceCountD>('tsh_device_count');
      _lastError = lib.lookupFunction<_LastErrorC, _LastErrorD>('tsh_last_error');
      _create = lib.lookupFunction<_CreateC, _CreateD>('tsh_index_create');
      _destroy = lib.lookupFunction<_DestroyC, _DestroyD>('tsh_index_destroy');
      _append = lib.lookupFunction<_AppendC, _AppendD>('tsh_index_append');
      _setDeleted = lib.lookupFunction<_SetDeletedC, _SetDeletedD>('tsh_index_set_deleted');
      _loadRawvec =
          lib.lookupFunction<_LoadRawvecC, _LoadRawvecD>('tsh_index_load_rawvec_file');
      _size = lib.lookupFunction<_SizeC, _SizeD>('tsh_index_size');
      _search = lib.lookupFunction<_SearchC, _SearchD>('tsh_search');
      _openNgh = lib.lookupFunction<_OpenNghC, _OpenNghD>('tsh_index_open_ngh');
      _pqEncode = lib.lookupFunction<_PqEncodeC, _PqEncodeD>('tsh_index_pq_encode');
      _pqTrain = lib.lookupFunction<_PqTrainC, _PqTrainD>('tsh_pq_train');
      if (_abiVersion() != 1 || _deviceCount() < 1) return false;
      _lib = lib;
      return true;
    } catch (_) {
      return false; // same catch-and-fall-back style as SystemFFIHelper
    }
  }

  static String _errorText() {
    final buf = calloc<Uint8>(512);
    try {
      _lastError(buf.cast<Utf8>(), 512);
      return buf.cast<Utf8>().toDartString();
    } finally {
      calloc.free(buf);
    }
  }

  Pointer<Void> _handle;
  final int dimensions;
  final VectorDistanceMetric metric;

  HipVectorBackend._(this._handle, this.dimensions, this.metric);

  /// metric index = enum order of VectorDistanceMetric (table_schema.dart:2511-2531).
  static HipVectorBackend? tryCreate(NghIndexMeta meta, {int devices = 1}) {
    if (!available) return null;
    final out = calloc<Pointer<Void>>();
    try {
      final rc = _create(meta.dimensions, meta.distanceMetric.index,
          meta.nextNodeId > 0 ? meta.nextNodeId : 0, devices, out);
      if (rc != 0) {
        Logger.warn('tsh_index_create failed ($rc): ${_errorText()}',
            label: 'HipVectorBackend');
        return null;
      }
      return HipVectorBackend._(out.value, meta.dimensions, meta.distanceMetric);
    } finally {
      calloc.free(out);
    }
  }

  /// Cold start straight from `<index>/ngh` (path_manager.dart:275-278): the library reads
  /// meta.json, every rawvec partition and the graph slots' deleted flags itself.
  static HipVectorBackend? tryOpen(String nghDir, NghIndexMeta meta, int maxEntriesPerDir,
      {int devices = 1}) {
    if (!available) return null;
    final out = calloc<Pointer<Void>>();
    final info = calloc<TshNghInfo>();
    final p = nghDir.toNativeUtf8();
    try {
      final rc = _openNgh(p, maxEntriesPerDir, devices, out, info);
      if (rc != 0) {
        Logger.warn('tsh_index_open_ngh failed ($rc): ${_errorText()}',
            label: 'HipVectorBackend');
        return null;
      }
      // Raw-vector pages that are not on disk load as ABSENT rows: the exhaustive path would
      // silently never return those nodes, so an index with holes stays on the Dart path.
      if (info.ref.pagesAbsent != 0 || info.ref.rowsLoaded != meta.nextNodeId) {
        Logger.warn(
            'device index of $nghDir not used: ${info.ref.pagesAbsent} raw-vector pages absent, '
            '${info.ref.rowsLoaded} of ${meta.nextNodeId} rows loaded',
            label: 'HipVectorBackend');
        _destroy(out.value);
        return null;
      }
      return HipVectorBackend._(out.value, meta.dimensions, meta.distanceMetric);
    } finally {
      calloc.free(p);
      calloc.free(info);
      calloc.free(out);
    }
  }

  int get size => _size(_handle);

  /// Replaces the trainPqSubspace isolate fan-out of _ensurePqCodebook
  /// (vector_index_manager.dart:740-850): same codebook, bit for bit, because the
  /// seeds are drawn here exactly as compute_tasks.dart:2144-2151 draws them.
  static Float32List? trainCodebook(List<Float32List> samples, int dimensions, int subspaces,
      {int iterations = 10}) {
    if (!available || samples.length < 100) return null; // small sets keep VectorQuantizer.train
    final n = samples.length;
    final k = n < 256 ? n : 256;
    final subDim = dimensions ~/ subspaces;
    final pSamples = calloc<Float>(n * dimensions);
    final pInit = calloc<Int32>(subspaces * k);
    final pOut = calloc<Float>(subspaces * k * subDim);
    try {
      final flat = pSamples.asTypedList(n * dimensions);
      for (int i = 0; i < n; i++) {
        flat.setRange(i * dimensions, (i + 1) * dimensions, samples[i]);
      }
      final init = pInit.asTypedList(subspaces * k);
      for (int m = 0; m < subspaces; m++) {
        final random = Random(42 + m);
        for (int c = 0; c < k; c++) {
          init[m * k + c] = random.nextInt(n);
        }
      }
      final rc = _pqTrain(0, pSamples, n, dimensions, subspaces, k, iterations, pInit, pOut);
      if (rc != 0) {
        Logger.warn('tsh_pq_train failed ($rc): ${_errorText()}', label: 'HipVectorBackend');
        return null;
      }
      return Float32List.fromList(pOut.asTypedList(subspaces * k * subDim));
    } finally {
      calloc.free(pSamples);
      calloc.free(pInit);
      calloc.free(pOut);
    }
  }

  /// Replaces batchPqEncode (compute_tasks.dart:2292-2326) for rows that are already
  /// resident: n x subspaces code bytes for node ids [firstNodeId, firstNodeId + n).
  Uint8List? pqEncode(int firstNodeId, int n, PqCodebook codebook) {
    final pCb = calloc<Float>(codebook.data.length);
    final pOut = calloc<Uint8>(n * codebook.subspaces);
    try {
      pCb.asTypedList(codebook.data.length).setAll(0, codebook.data);
      final rc = _pqEncode(_handle, firstNodeId, n, pCb, codebook.subspaces,
          codebook.centroids, pOut);
      if (rc != 0) {
        Logger.warn('tsh_index_pq_encode failed ($rc): ${_errorText()}',
            label: 'HipVectorBackend');
        return null;
      }
      return Uint8List.fromList(pOut.asTypedList(n * codebook.subspaces));
    } finally {
      calloc.free(pCb);
      calloc.free(pOut);
    }
  }

  /// Feed from NghGraphEngine.insertBatch (ngh_graph_engine.dart:297-403): the
  /// Float32Lists produced by prepareVectorBatchChunk, ids dense from `firstNodeId`.
  bool append(int firstNodeId, List<Float32List> vectors) {
    if (vectors.isEmpty) return true;
    final n = vectors.length;
    final buf = calloc<Float>(n * dimensions);
    try {
      final view = buf.asTypedList(n * dimensions);
      for (var i = 0; i < n; i++) {
        view.setRange(i * dimensions, (i + 1) * dimensions, vectors[i]);
      }
      final rc = _append(_handle, firstNodeId, n, buf);
      if (rc != 0) Logger.warn('tsh_index_append failed ($rc): ${_errorText()}');
      return rc == 0;
    } finally {
      calloc.free(buf);
    }
  }

  /// Feed from NghGraphEngine.deleteBatch (ngh_graph_engine.dart:411-445).
  bool setDeleted(List<int> nodeIds) {
    if (nodeIds.isEmpty) return true;
    final buf = calloc<Int64>(nodeIds.length);
    try {
      buf.asTypedList(nodeIds.length).setAll(0, nodeIds);
      return _setDeleted(_handle, buf, nodeIds.length) == 0;
    } finally {
      calloc.free(buf);
    }
  }

  /// Cold load of one rawvec partition file (path_manager.dart:318-324).
  int loadRawVectorFile(String path, NghIndexMeta meta, int firstNodeId, int maxRows) {
    final p = path.toNativeUtf8();
    final out = calloc<Int64>();
    try {
      final rc = _loadRawvec(_handle, p, meta.nghPageSize, meta.precision.index,
          firstNodeId, maxRows, out);
      return rc == 0 ? out.value : -1;
    } finally {
      calloc.free(p);
      calloc.free(out);
    }
  }

  /// Drop-in for NghGraphEngine.search (ngh_graph_engine.dart:67-135).  `query`
  /// is already _toFloat32'ed and (cosine) _normalizeFloat32'ed by the caller
  /// (vector_index_manager.dart:514-520).  Returns null on any native failure so
  /// the caller falls through to the original graph search.
  List<NghSearchResult>? search(Float32List query, int topK,
      {double? distanceThreshold, Uint8List? rowMask}) {
    if (topK <= 0 || size == 0) return const [];
    final q = calloc<Float>(dimensions);
    final ids = calloc<Int64>(topK);
    final dist = calloc<Double>(topK);
    final cnt = calloc<Int32>();
    Pointer<Uint8> mask = nullptr;
    try {
      q.asTypedList(dimensions).setAll(0, query);
      if (rowMask != null) {
        mask = calloc<Uint8>(rowMask.length);
        mask.asTypedList(rowMask.length).setAll(0, rowMask);
      }
      final rc = _search(_handle, q, 1, topK, distanceThreshold ?? double.nan, mask,
          ids, dist, cnt);
      if (rc != 0) {
        Logger.warn('tsh_search failed ($rc): ${_errorText()}', label: 'HipVectorBackend');
        return null;
      }
      final n = cnt.value;
      return [
        for (var i = 0; i < n; i++) NghSearchResult(nodeId: ids[i], distance: dist[i])
      ];
    } finally {
      calloc.free(q);
      calloc.free(ids);
      calloc.free(dist);
      calloc.free(cnt);
      if (mask != nullptr) calloc.free(mask);
    }
  }

  void dispose() {
    if (_handle != nullptr) {
      _destroy(_handle);
      _handle = nullptr;
    }
  }
}
