// tostore_hip_bridge.dart -- the `dart:ffi` binding a ToStore maintainer adds as
// lib/src/handler/hip_vector_backend.dart to route vectorSearch() through
// libtostore_hip.so (include/tostore_hip.h).
//
// NOT compiled in this repository: the build image has no Dart SDK.  What CAN be
// checked without one is: tests/test_dart_bridge.py parses every `typedef ...C`
// below, the symbol each one is looked up as, and the TshNghInfo / TshCounters
// structs, and compares arity, argument widths and field layout with
// include/tostore_hip.h; a header symbol that is neither bound here nor on that
// test's short "not for a Dart host" list fails the test.
// It is a mechanical mapping of the C-ABI, written in the style of the
// reference's only existing FFI user, lib/src/handler/system_ffi_helper.dart
// (DynamicLibrary.open + lookupFunction, int32 status with 0 = success,
// calloc/free in try/finally, every failure falls back to the Dart path).
// The ctypes binding tostore_amd/_ffi.py exercises the same entry points with
// the same argument meaning and IS tested (tests/test_abi.py, tests/test_gpu_*).

import 'dart:async';
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
typedef _OpenNghShardC = Int32 Function(
    Pointer<Utf8>, Int32, Int32, Int32, Int32, Pointer<Pointer<Void>>, Pointer<TshNghInfo>);
typedef _OpenNghShardD = int Function(
    Pointer<Utf8>, int, int, int, int, Pointer<Pointer<Void>>, Pointer<TshNghInfo>);
typedef _MaskCreateC = Int32 Function(Pointer<Void>, Pointer<Uint8>, Int64, Pointer<Pointer<Void>>);
typedef _MaskCreateD = int Function(Pointer<Void>, Pointer<Uint8>, int, Pointer<Pointer<Void>>);
typedef _MaskDestroyC = Int32 Function(Pointer<Void>);
typedef _MaskDestroyD = int Function(Pointer<Void>);
typedef _MaskKeptC = Int64 Function(Pointer<Void>);
typedef _MaskKeptD = int Function(Pointer<Void>);
typedef _SearchMaskedC = Int32 Function(Pointer<Void>, Pointer<Float>, Int32, Int32, Double,
    Pointer<Void>, Pointer<Int64>, Pointer<Double>, Pointer<Int32>);
typedef _SearchMaskedD = int Function(Pointer<Void>, Pointer<Float>, int, int, double,
    Pointer<Void>, Pointer<Int64>, Pointer<Double>, Pointer<Int32>);
typedef _SubmitMaskedC = Int32 Function(Pointer<Void>, Pointer<Float>, Int32, Pointer<Void>, Pointer<Int32>);
typedef _SubmitMaskedD = int Function(Pointer<Void>, Pointer<Float>, int, Pointer<Void>, Pointer<Int32>);
typedef _PqEncodeC = Int32 Function(
    Pointer<Void>, Int64, Int64, Pointer<Float>, Int32, Int32, Pointer<Uint8>);
typedef _PqEncodeD = int Function(
    Pointer<Void>, int, int, Pointer<Float>, int, int, Pointer<Uint8>);
typedef _PqTrainC = Int32 Function(Int32, Pointer<Float>, Int64, Int32, Int32, Int32, Int32,
    Pointer<Int32>, Pointer<Float>);
typedef _PqTrainD = int Function(int, Pointer<Float>, int, int, int, int, int,
    Pointer<Int32>, Pointer<Float>);

typedef _CreateShardC = Int32 Function(Int32, Int32, Int64, Int32, Int64, Pointer<Pointer<Void>>);
typedef _CreateShardD = int Function(int, int, int, int, int, Pointer<Pointer<Void>>);
typedef _DimC = Int32 Function(Pointer<Void>);
typedef _DimD = int Function(Pointer<Void>);
typedef _MetricC = Int32 Function(Pointer<Void>);
typedef _MetricD = int Function(Pointer<Void>);
typedef _MaxInflightC = Int32 Function();
typedef _MaxInflightD = int Function();
typedef _SubmitC = Int32 Function(Pointer<Void>, Pointer<Float>, Int32, Pointer<Uint8>, Pointer<Int32>);
typedef _SubmitD = int Function(Pointer<Void>, Pointer<Float>, int, Pointer<Uint8>, Pointer<Int32>);
typedef _ReadyC = Int32 Function(Pointer<Void>, Int32);
typedef _ReadyD = int Function(Pointer<Void>, int);
typedef _WaitC = Int32 Function(
    Pointer<Void>, Int32, Double, Pointer<Int64>, Pointer<Double>, Pointer<Int32>);
typedef _WaitD = int Function(
    Pointer<Void>, int, double, Pointer<Int64>, Pointer<Double>, Pointer<Int32>);
typedef _CountersC = Int32 Function(Pointer<Void>, Pointer<TshCounters>);
typedef _CountersD = int Function(Pointer<Void>, Pointer<TshCounters>);
typedef _SetOptionC = Int32 Function(Pointer<Void>, Int32, Int64);
typedef _SetOptionD = int Function(Pointer<Void>, int, int);
typedef _BlockBytesC = Int64 Function(Int32);
typedef _BlockBytesD = int Function(int);
typedef _BlockEntriesC = Int32 Function(Int32);
typedef _BlockEntriesD = int Function(int);
typedef _SearchShardC = Int32 Function(Pointer<Void>, Pointer<Float>, Int32, Int32, Pointer<Uint8>,
    Int32, Pointer<Void>, Pointer<Void>);
typedef _SearchShardD = int Function(Pointer<Void>, Pointer<Float>, int, int, Pointer<Uint8>,
    int, Pointer<Void>, Pointer<Void>);
typedef _ShardBeginC = Int32 Function(Pointer<Void>, Pointer<Float>, Int32, Int32, Pointer<Uint8>,
    Int32, Pointer<Void>, Int32, Pointer<Pointer<Void>>);
typedef _ShardBeginD = int Function(Pointer<Void>, Pointer<Float>, int, int, Pointer<Uint8>,
    int, Pointer<Void>, int, Pointer<Pointer<Void>>);
typedef _ShardProgressC = Int32 Function(Pointer<Void>, Int32, Pointer<Int32>);
typedef _ShardProgressD = int Function(Pointer<Void>, int, Pointer<Int32>);
typedef _ShardEndC = Int32 Function(Pointer<Void>);
typedef _ShardEndD = int Function(Pointer<Void>);
typedef _MergeC = Int32 Function(Int32, Int32, Pointer<Float>, Int32, Int32, Double, Pointer<Void>,
    Int32, Int32, Pointer<Int64>, Pointer<Double>, Pointer<Int32>, Pointer<Int32>);
typedef _MergeD = int Function(int, int, Pointer<Float>, int, int, double, Pointer<Void>,
    int, int, Pointer<Int64>, Pointer<Double>, Pointer<Int32>, Pointer<Int32>);
typedef _CommIdC = Int32 Function(Pointer<Void>);
typedef _CommIdD = int Function(Pointer<Void>);
typedef _CommCreateC = Int32 Function(Pointer<Void>, Int32, Int32, Int32, Pointer<Pointer<Void>>);
typedef _CommCreateD = int Function(Pointer<Void>, int, int, int, Pointer<Pointer<Void>>);
/// `tsh_allgather_fn`: int32 (*)(void *user, const void *send, void *recv, int64 bytes)
typedef TshAllgatherNative = Int32 Function(Pointer<Void>, Pointer<Void>, Pointer<Void>, Int64);
typedef _CommCreateHostC = Int32 Function(Int32, Int32, Int32,
    Pointer<NativeFunction<TshAllgatherNative>>, Pointer<Void>, Pointer<Pointer<Void>>);
typedef _CommCreateHostD = int Function(int, int, int,
    Pointer<NativeFunction<TshAllgatherNative>>, Pointer<Void>, Pointer<Pointer<Void>>);
typedef _CommDestroyC = Int32 Function(Pointer<Void>);
typedef _CommDestroyD = int Function(Pointer<Void>);
typedef _CommWorldC = Int32 Function(Pointer<Void>);
typedef _CommWorldD = int Function(Pointer<Void>);
typedef _CommSetGroupC = Int32 Function(Pointer<Void>, Int32);
typedef _CommSetGroupD = int Function(Pointer<Void>, int);
typedef _CommTimelineC = Int32 Function(Pointer<Void>, Pointer<TshCommTimeline>, Int32);
typedef _CommTimelineD = int Function(Pointer<Void>, Pointer<TshCommTimeline>, int);
typedef _SearchShardedC = Int32 Function(Pointer<Void>, Pointer<Void>, Pointer<Float>, Int32, Int32,
    Double, Pointer<Uint8>, Pointer<Int64>, Pointer<Double>, Pointer<Int32>);
typedef _SearchShardedD = int Function(Pointer<Void>, Pointer<Void>, Pointer<Float>, int, int,
    double, Pointer<Uint8>, Pointer<Int64>, Pointer<Double>, Pointer<Int32>);

/// `tsh_counters` (include/tostore_hip.h).  Field order and widths are checked against the header by
/// tests/test_dart_bridge.py.
final class TshCounters extends Struct {
  @Int64()
  external int rows;
  @Int64()
  external int deletedRows;
  @Int64()
  external int searches;
  @Int64()
  external int scanLaunches;
  @Int64()
  external int batchLaunches;
  @Int64()
  external int fallbackSearches;
  @Int64()
  external int candidatesTotal;
  @Int64()
  external int bytesResident;
  @Int32()
  external int safeMode;
  @Int32()
  external int deviceId;
  @Double()
  external double scanUsSum;
  @Int64()
  external int scanUsSamples;
  @Int32()
  external int batchKernelLast;
  @Int32()
  external int quarantinedRows;
  @Int64()
  external int exactScans;
  @Int64()
  external int batchPlaneFallbacks;
  @Int64()
  external int batchScanFallbacks;
  @Int64()
  external int listScans;
  @Int64()
  external int exactRedone;
}

/// `tsh_comm_timeline` (include/tostore_hip.h): where this rank's tsh_search_sharded time went.  Field order and
/// widths are checked against the header by tests/test_dart_bridge.py.
final class TshCommTimeline extends Struct {
  @Int64()
  external int calls;
  @Int64()
  external int queries;
  @Int64()
  external int groups;
  @Int64()
  external int retries;
  @Int32()
  external int world;
  @Int32()
  external int rank;
  @Int32()
  external int transport;
  @Int32()
  external int reserved;
  @Double()
  external double callUs;
  @Double()
  external double reserveUs;
  @Double()
  external double waitScanUs;
  @Double()
  external double scanUs;
  @Double()
  external double exchangeWaitUs;
  @Double()
  external double gatherUs;
  @Double()
  external double sliceD2hUs;
  @Double()
  external double mergeUs;
  @Double()
  external double resultGatherUs;
  @Double()
  external double copyOutUs;
  @Double()
  external double retryScanUs;
  @Double()
  external double preEnqueueUs;
}

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
  @Int64()
  external int rowBase;
  @Int64()
  external int rowEnd;
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
  static late final _CreateShardD _createShard;
  static late final _DimD _dim;
  static late final _MetricD _metric;
  static late final _MaxInflightD _maxInflight;
  static late final _SubmitD _submit;
  static late final _ReadyD _ready;
  static late final _WaitD _wait;
  static late final _CountersD _counters;
  static late final _SetOptionD _setOption;
  static late final _BlockBytesD _blockBytes;
  static late final _BlockEntriesD _blockEntries;
  static late final _SearchShardD _searchShard;
  static late final _ShardBeginD _shardBegin;
  static late final _ShardProgressD _shardProgress;
  static late final _ShardEndD _shardEnd;
  static late final _MergeD _merge;
  static late final _CommIdD _commId;
  static late final _CommCreateD _commCreate;
  static late final _CommCreateHostD _commCreateHost;
  static late final _CommDestroyD _commDestroy;
  static late final _CommWorldD _commWorld;
  static late final _CommSetGroupD _commSetGroup;
  static late final _SearchShardedD _searchSharded;
  static late final _CommTimelineD _commTimeline;
  static late final _OpenNghShardD _openNghShard;
  static late final _MaskCreateD _maskCreate;
  static late final _MaskDestroyD _maskDestroy;
  static late final _MaskKeptD _maskKept;
  static late final _SearchMaskedD _searchMasked;
  static late final _SubmitMaskedD _submitMasked;

  /// include/tostore_hip.h TSH_ABI_VERSION this file was written against.
  static const int abiVersion = 5;

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
      _createShard = lib.lookupFunction<_CreateShardC, _CreateShardD>('tsh_index_create_shard');
      _dim = lib.lookupFunction<_DimC, _DimD>('tsh_index_dim');
      _metric = lib.lookupFunction<_MetricC, _MetricD>('tsh_index_metric');
      _maxInflight = lib.lookupFunction<_MaxInflightC, _MaxInflightD>('tsh_max_inflight');
      _submit = lib.lookupFunction<_SubmitC, _SubmitD>('tsh_search_submit');
      _ready = lib.lookupFunction<_ReadyC, _ReadyD>('tsh_search_ready');
      _wait = lib.lookupFunction<_WaitC, _WaitD>('tsh_search_wait');
      _counters = lib.lookupFunction<_CountersC, _CountersD>('tsh_get_counters');
      _setOption = lib.lookupFunction<_SetOptionC, _SetOptionD>('tsh_index_set_option');
      _blockBytes = lib.lookupFunction<_BlockBytesC, _BlockBytesD>('tsh_candidate_block_bytes');
      _blockEntries = lib.lookupFunction<_BlockEntriesC, _BlockEntriesD>('tsh_default_block_entries');
      _searchShard = lib.lookupFunction<_SearchShardC, _SearchShardD>('tsh_search_shard');
      _shardBegin = lib.lookupFunction<_ShardBeginC, _ShardBeginD>('tsh_search_shard_begin');
      _shardProgress =
          lib.lookupFunction<_ShardProgressC, _ShardProgressD>('tsh_search_shard_progress');
      _shardEnd = lib.lookupFunction<_ShardEndC, _ShardEndD>('tsh_search_shard_end');
      _merge = lib.lookupFunction<_MergeC, _MergeD>('tsh_merge_candidates');
      _commId = lib.lookupFunction<_CommIdC, _CommIdD>('tsh_comm_unique_id');
      _commCreate = lib.lookupFunction<_CommCreateC, _CommCreateD>('tsh_comm_create');
      _commCreateHost =
          lib.lookupFunction<_CommCreateHostC, _CommCreateHostD>('tsh_comm_create_host');
      _commDestroy = lib.lookupFunction<_CommDestroyC, _CommDestroyD>('tsh_comm_destroy');
      _commWorld = lib.lookupFunction<_CommWorldC, _CommWorldD>('tsh_comm_world');
      _commSetGroup = lib.lookupFunction<_CommSetGroupC, _CommSetGroupD>('tsh_comm_set_group');
      _searchSharded =
          lib.lookupFunction<_SearchShardedC, _SearchShardedD>('tsh_search_sharded');
      _commTimeline =
          lib.lookupFunction<_CommTimelineC, _CommTimelineD>('tsh_comm_get_timeline');
      _openNghShard =
          lib.lookupFunction<_OpenNghShardC, _OpenNghShardD>('tsh_index_open_ngh_shard');
      _maskCreate = lib.lookupFunction<_MaskCreateC, _MaskCreateD>('tsh_mask_create');
      _maskDestroy = lib.lookupFunction<_MaskDestroyC, _MaskDestroyD>('tsh_mask_destroy');
      _maskKept = lib.lookupFunction<_MaskKeptC, _MaskKeptD>('tsh_mask_kept');
      _searchMasked = lib.lookupFunction<_SearchMaskedC, _SearchMaskedD>('tsh_search_masked');
      _submitMasked =
          lib.lookupFunction<_SubmitMaskedC, _SubmitMaskedD>('tsh_search_submit_masked');
      // the structs of this file are the version-5 layouts: any other library is not used
      if (_abiVersion() != abiVersion || _deviceCount() < 1) return false;
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

  /// Cold start of ONE rank's row range of `<index>/ngh` (one Dart process per GPU, [HipShardComm]): node ids
  /// [rank * ceil(nextNodeId / world), ...) as a shard handle with global ids -- the library opens only the
  /// partition files and reads only the pages that hold the range (model/ngh_index_meta.dart:480-490,
  /// core/path_manager.dart:275-324).  Null when the range has holes on disk, like [tryOpen].
  static HipVectorBackend? tryOpenShard(String nghDir, NghIndexMeta meta, int maxEntriesPerDir,
      int device, int world, int rank) {
    if (!available) return null;
    final out = calloc<Pointer<Void>>();
    final info = calloc<TshNghInfo>();
    final p = nghDir.toNativeUtf8();
    try {
      final rc = _openNghShard(p, maxEntriesPerDir, device, world, rank, out, info);
      if (rc != 0) {
        Logger.warn('tsh_index_open_ngh_shard failed ($rc): ${_errorText()}',
            label: 'HipVectorBackend');
        return null;
      }
      final want = info.ref.rowEnd - info.ref.rowBase;
      if (info.ref.pagesAbsent != 0 || info.ref.rowsLoaded != want) {
        Logger.warn(
            'device shard $rank/$world of $nghDir not used: ${info.ref.pagesAbsent} raw-vector pages '
            'absent, ${info.ref.rowsLoaded} of $want rows loaded',
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

  /// A WHERE row set that lives on the device across queries (tsh_mask_create): bit i of `bits` (LSB first)
  /// keeps node id i.  The bitmap crosses the FFI boundary ONCE; searches that pass the handle
  /// ([search] / [searchAsync] `mask:`) copy nothing and do no host work on it -- the `rowMask:` form
  /// callocs, copies and has the library slice, count and list the bitmap on every call.  Dispose the mask
  /// before this backend.
  HipRowMask? createMask(Uint8List bits) {
    final buf = calloc<Uint8>(bits.isEmpty ? 1 : bits.length);
    final out = calloc<Pointer<Void>>();
    try {
      buf.asTypedList(bits.length).setAll(0, bits);
      final rc = _maskCreate(_handle, buf, bits.length, out);
      if (rc != 0) {
        Logger.warn('tsh_mask_create failed ($rc): ${_errorText()}', label: 'HipVectorBackend');
        return null;
      }
      return HipRowMask._(out.value);
    } finally {
      calloc.free(buf);
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
      {double? distanceThreshold, Uint8List? rowMask, HipRowMask? mask}) {
    if (topK <= 0 || size == 0) return const [];
    final q = calloc<Float>(dimensions);
    final ids = calloc<Int64>(topK);
    final dist = calloc<Double>(topK);
    final cnt = calloc<Int32>();
    Pointer<Uint8> maskBytes = nullptr;
    try {
      q.asTypedList(dimensions).setAll(0, query);
      if (mask == null && rowMask != null) {
        maskBytes = calloc<Uint8>(rowMask.length);
        maskBytes.asTypedList(rowMask.length).setAll(0, rowMask);
      }
      final rc = mask != null
          ? _searchMasked(_handle, q, 1, topK, distanceThreshold ?? double.nan, mask._mask,
              ids, dist, cnt)
          : _search(_handle, q, 1, topK, distanceThreshold ?? double.nan, maskBytes,
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
      if (maskBytes != nullptr) calloc.free(maskBytes);
    }
  }

  /// One shard of a row-range partitioned index (one Dart process per GPU): holds node ids
  /// [rowBase, rowBase + rows) on `device`; reported ids are global.
  static HipVectorBackend? tryCreateShard(NghIndexMeta meta, int device, int rowBase, int rows) {
    if (!available) return null;
    final out = calloc<Pointer<Void>>();
    try {
      final rc = _createShard(
          meta.dimensions, meta.distanceMetric.index, rows, device, rowBase, out);
      if (rc != 0) {
        Logger.warn('tsh_index_create_shard failed ($rc): ${_errorText()}',
            label: 'HipVectorBackend');
        return null;
      }
      return HipVectorBackend._(out.value, meta.dimensions, meta.distanceMetric);
    } finally {
      calloc.free(out);
    }
  }

  /// Searches that may be in flight per handle (tsh_max_inflight).
  static int get maxInflight => available ? _maxInflight() : 0;

  /// The asynchronous form of [search]: the query is enqueued (some 20 microseconds), the isolate goes back
  /// to its event loop while the GPU scans, and the result is collected once tsh_search_ready says
  /// so -- the isolate never blocks for longer than the final copy, whatever the corpus size, which
  /// is what the reference's cooperative scheduling asks of everything on the main isolate (8 ms
  /// client / 50 ms server budget: model/data_store_config.dart:225-230,
  /// core/yield_controller.dart:110-169).  Up to [maxInflight] calls may overlap on one handle; their
  /// scans run back to back and each query's select / re-rank hides behind the next scan.
  /// Returns null on any native failure (TSH_E_BUSY included: the caller falls back to the
  /// synchronous path or the Dart graph search).
  Future<List<NghSearchResult>?> searchAsync(Float32List query, int topK,
      {double? distanceThreshold, Uint8List? rowMask, HipRowMask? mask}) async {
    if (topK <= 0 || size == 0) return const [];
    final q = calloc<Float>(dimensions);
    final ticket = calloc<Int32>();
    Pointer<Uint8> maskBytes = nullptr;
    int t;
    try {
      q.asTypedList(dimensions).setAll(0, query);
      if (mask == null && rowMask != null) {
        maskBytes = calloc<Uint8>(rowMask.length);
        maskBytes.asTypedList(rowMask.length).setAll(0, rowMask);
      }
      // inputs are consumed before either returns (a mask HANDLE must outlive the ticket)
      final rc = mask != null
          ? _submitMasked(_handle, q, topK, mask._mask, ticket)
          : _submit(_handle, q, topK, maskBytes, ticket);
      if (rc != 0) {
        Logger.warn('tsh_search_submit failed ($rc): ${_errorText()}',
            label: 'HipVectorBackend');
        return null;
      }
      t = ticket.value;
    } finally {
      calloc.free(q);
      calloc.free(ticket);
      if (maskBytes != nullptr) calloc.free(maskBytes);
    }
    // every ticket must be waited exactly once: from here on nothing may return before _wait ran.
    // The first polls yield with a zero delay (a short scan is over by then); after that the isolate sleeps
    // between polls -- a zero-delay timer loop would keep its event loop at 100 % CPU for the whole scan, and
    // several isolates or ranks per container share one CPU quota.
    var polls = 0;
    while (true) {
      final ready = _ready(_handle, t);
      if (ready > 0) break;
      if (ready < 0) {
        // the ticket is not pollable (it still must be waited: _wait reports the error and releases it)
        Logger.warn('tsh_search_ready failed ($ready): ${_errorText()}', label: 'HipVectorBackend');
        break;
      }
      polls++;
      await Future<void>.delayed(
          polls <= 4 ? Duration.zero : Duration(microseconds: polls <= 12 ? 50 : 200));
    }
    final ids = calloc<Int64>(topK);
    final dist = calloc<Double>(topK);
    final cnt = calloc<Int32>();
    try {
      final rc = _wait(_handle, t, distanceThreshold ?? double.nan, ids, dist, cnt);
      if (rc != 0) {
        Logger.warn('tsh_search_wait failed ($rc): ${_errorText()}', label: 'HipVectorBackend');
        return null;
      }
      final n = cnt.value;
      return [
        for (var i = 0; i < n; i++) NghSearchResult(nodeId: ids[i], distance: dist[i])
      ];
    } finally {
      calloc.free(ids);
      calloc.free(dist);
      calloc.free(cnt);
    }
  }

  /// Diagnostics for Logger (handler/logger.dart:8-60): rows, searches, candidates per query,
  /// resident bytes, safe mode / quarantined rows.
  Map<String, num>? counters() {
    final c = calloc<TshCounters>();
    try {
      if (_counters(_handle, c) != 0) return null;
      final r = c.ref;
      return {
        'rows': r.rows,
        'deletedRows': r.deletedRows,
        'searches': r.searches,
        'scanLaunches': r.scanLaunches,
        'batchLaunches': r.batchLaunches,
        'fallbackSearches': r.fallbackSearches,
        'candidatesTotal': r.candidatesTotal,
        'bytesResident': r.bytesResident,
        'safeMode': r.safeMode,
        'deviceId': r.deviceId,
        'quarantinedRows': r.quarantinedRows,
        'batchKernelLast': r.batchKernelLast,
        'batchPlaneFallbacks': r.batchPlaneFallbacks,
        'batchScanFallbacks': r.batchScanFallbacks,
        'listScans': r.listScans,
        'exactScans': r.exactScans,
        'exactRedone': r.exactRedone,
      };
    } finally {
      calloc.free(c);
    }
  }

  /// tsh_index_set_option: 1 = TSH_OPT_BATCH_MIN_NQ, 2 = TSH_OPT_BATCH_KERNEL, 4 = TSH_OPT_EXACT_SCAN_ROWS,
  /// 5 = TSH_OPT_EXACT_SELECT, 6 = TSH_OPT_BATCH_HUB, 7 = TSH_OPT_BATCH_GROUP (tuning only: results never depend on them).
  bool setOption(int option, int value) => _setOption(_handle, option, value) == 0;

  int get nativeDimensions => _dim(_handle);
  int get nativeMetric => _metric(_handle);

  /// Row-sharded deployments whose exchange the HOST does itself (any transport): this shard's
  /// candidate blocks for `queries` into `deviceBlocks` (device memory of
  /// nq * candidateBlockBytes(entries) bytes), to be gathered from all ranks and merged with
  /// [mergeCandidates].  Hosts on one node use [HipShardComm] instead, which does all of it.
  static int candidateBlockBytes(int entries) => _blockBytes(entries);
  static int defaultBlockEntries(int k) => _blockEntries(k);
  bool searchShard(Pointer<Float> queries, int nq, int topK, Pointer<Uint8> rowMask, int entries,
          Pointer<Void> deviceBlocks) =>
      _searchShard(_handle, queries, nq, topK, rowMask, entries, deviceBlocks, nullptr) == 0;

  /// Progressive form (tsh_search_shard_begin / _progress / _end): the scans of all `nq` queries run as one
  /// pipeline on a library thread while the host exchanges the groups whose blocks are final.  Returns the
  /// stream handle (nullptr on failure); `step` = the host's group size.  The queries / mask are copied by the
  /// call; `deviceBlocks` must stay valid until [shardStreamEnd].
  Pointer<Void> shardStreamBegin(Pointer<Float> queries, int nq, int topK, Pointer<Uint8> rowMask,
      int entries, Pointer<Void> deviceBlocks, int step) {
    final out = calloc<Pointer<Void>>();
    try {
      final rc = _shardBegin(_handle, queries, nq, topK, rowMask, entries, deviceBlocks, step, out);
      return rc == 0 ? out.value : nullptr;
    } finally {
      calloc.free(out);
    }
  }

  /// Blocks until the first min(want, nq) queries' blocks are final; -1 when the search failed first,
  /// otherwise the number of leading queries that are final.
  static int shardStreamProgress(Pointer<Void> stream, int want) {
    final done = calloc<Int32>();
    try {
      return _shardProgress(stream, want, done) == 0 ? done.value : -1;
    } finally {
      calloc.free(done);
    }
  }

  /// Waits for whatever still runs and frees the stream; exactly once per [shardStreamBegin].
  static bool shardStreamEnd(Pointer<Void> stream) => _shardEnd(stream) == 0;

  /// Host-side merge of `nBlocks` x nq gathered candidate blocks; false with `neededEntries`
  /// set when a block was truncated (every rank retries with that many entries).
  static bool mergeCandidates(int metric, int dim, Pointer<Float> queries, int nq, int topK,
      double? distanceThreshold, Pointer<Void> blocks, int nBlocks, int entries,
      Pointer<Int64> outIds, Pointer<Double> outDist, Pointer<Int32> outCount,
      Pointer<Int32> neededEntries) {
    return _merge(metric, dim, queries, nq, topK, distanceThreshold ?? double.nan, blocks, nBlocks,
            entries, outIds, outDist, outCount, neededEntries) ==
        0;
  }

  void dispose() {
    if (_handle != nullptr) {
      _destroy(_handle);
      _handle = nullptr;
    }
  }
}

/// One rank of a row-sharded index: one Dart process per GPU, the library's own RCCL exchange
/// (tsh_comm_* / tsh_search_sharded).  Rank 0 calls [uniqueId] and ships the 128 bytes to its peers over
/// whatever channel the deployment has; every rank then constructs its communicator (collective) and
/// calls [search] with the same queries (collective).  Every rank gets the full answer.
final class HipShardComm {
  Pointer<Void> _comm;
  final HipVectorBackend shard;

  HipShardComm._(this._comm, this.shard);

  static Uint8List? uniqueId() {
    if (!HipVectorBackend.available) return null;
    final buf = calloc<Uint8>(128);
    try {
      if (HipVectorBackend._commId(buf.cast<Void>()) != 0) return null;
      return Uint8List.fromList(buf.asTypedList(128));
    } finally {
      calloc.free(buf);
    }
  }

  static HipShardComm? tryCreate(
      HipVectorBackend shard, Uint8List id, int world, int rank, int device) {
    if (!HipVectorBackend.available || id.length != 128) return null;
    final buf = calloc<Uint8>(128);
    final out = calloc<Pointer<Void>>();
    try {
      buf.asTypedList(128).setAll(0, id);
      final rc = HipVectorBackend._commCreate(buf.cast<Void>(), world, rank, device, out);
      if (rc != 0) {
        Logger.warn('tsh_comm_create failed ($rc): ${HipVectorBackend._errorText()}',
            label: 'HipShardComm');
        return null;
      }
      return HipShardComm._(out.value, shard);
    } finally {
      calloc.free(buf);
      calloc.free(out);
    }
  }

  /// The same protocol over a transport the host brings (ranks on several nodes): `allgather` is a
  /// `Pointer.fromFunction` / `NativeCallable.isolateLocal` of [TshAllgatherNative] that places every rank's
  /// bytes at recv + rank * bytes on every rank.
  static HipShardComm? tryCreateOverHost(HipVectorBackend shard, int world, int rank, int device,
      Pointer<NativeFunction<TshAllgatherNative>> allgather) {
    if (!HipVectorBackend.available) return null;
    final out = calloc<Pointer<Void>>();
    try {
      final rc =
          HipVectorBackend._commCreateHost(world, rank, device, allgather, nullptr, out);
      if (rc != 0) return null;
      return HipShardComm._(out.value, shard);
    } finally {
      calloc.free(out);
    }
  }

  int get world => HipVectorBackend._commWorld(_comm);

  /// Queries per exchange inside one [search] call (0 = by the size of the call).  Same on every rank.
  bool setGroup(int queriesPerExchange) =>
      HipVectorBackend._commSetGroup(_comm, queriesPerExchange) == 0;

  /// Collective: same queries, topK and threshold on every rank.  null on failure -- this rank's own
  /// (its error is logged) or another rank's (TSH_E_PEER = -11): every rank then takes the same
  /// fallback, and the communicator stays usable.
  List<List<NghSearchResult>>? search(List<Float32List> queries, int topK,
      {double? distanceThreshold, Uint8List? rowMask}) {
    final nq = queries.length, d = shard.dimensions;
    if (nq == 0 || topK <= 0) return [for (var i = 0; i < nq; i++) const []];
    final q = calloc<Float>(nq * d);
    final ids = calloc<Int64>(nq * topK);
    final dist = calloc<Double>(nq * topK);
    final cnt = calloc<Int32>(nq);
    Pointer<Uint8> mask = nullptr;
    try {
      final view = q.asTypedList(nq * d);
      for (var i = 0; i < nq; i++) {
        view.setRange(i * d, (i + 1) * d, queries[i]);
      }
      if (rowMask != null) {
        mask = calloc<Uint8>(rowMask.length);
        mask.asTypedList(rowMask.length).setAll(0, rowMask);
      }
      final rc = HipVectorBackend._searchSharded(shard._handle, _comm, q, nq, topK,
          distanceThreshold ?? double.nan, mask, ids, dist, cnt);
      if (rc != 0) {
        Logger.warn('tsh_search_sharded failed ($rc): ${HipVectorBackend._errorText()}',
            label: 'HipShardComm');
        return null;
      }
      return [
        for (var i = 0; i < nq; i++)
          [
            for (var j = 0; j < cnt[i]; j++)
              NghSearchResult(nodeId: ids[i * topK + j], distance: dist[i * topK + j])
          ]
      ];
    } finally {
      calloc.free(q);
      calloc.free(ids);
      calloc.free(dist);
      calloc.free(cnt);
      if (mask != nullptr) calloc.free(mask);
    }
  }

  /// Where this rank's [search] time went, summed over the calls so far (tsh_comm_timeline): microseconds per
  /// phase, for Logger diagnostics of a slow multi-GPU deployment.  `reset` starts a new period.
  Map<String, num>? timeline({bool reset = false}) {
    final t = calloc<TshCommTimeline>();
    try {
      if (HipVectorBackend._commTimeline(_comm, t, reset ? 1 : 0) != 0) return null;
      final r = t.ref;
      return {
        'calls': r.calls,
        'queries': r.queries,
        'groups': r.groups,
        'retries': r.retries,
        'world': r.world,
        'rank': r.rank,
        'transport': r.transport,
        'callUs': r.callUs,
        'reserveUs': r.reserveUs,
        'waitScanUs': r.waitScanUs,
        'scanUs': r.scanUs,
        'exchangeWaitUs': r.exchangeWaitUs,
        'gatherUs': r.gatherUs,
        'sliceD2hUs': r.sliceD2hUs,
        'mergeUs': r.mergeUs,
        'resultGatherUs': r.resultGatherUs,
        'copyOutUs': r.copyOutUs,
        'retryScanUs': r.retryScanUs,
        'preEnqueueUs': r.preEnqueueUs,
      };
    } finally {
      calloc.free(t);
    }
  }

  void dispose() {
    if (_comm != nullptr) {
      HipVectorBackend._commDestroy(_comm);
      _comm = nullptr;
    }
  }
}

/// A device-resident WHERE row set ([HipVectorBackend.createMask], tsh_mask_create): the rows a structured
/// filter keeps -- primary keys mapped to node ids through the pk -> nodeId tree
/// (vector_index_manager.dart:1223-1378) -- or the complement of a tombstone set (ngh_page.dart:105-108),
/// kept for as many queries as it serves.  Rows appended after it was made are not kept; rows deleted
/// later are dropped by the kernels as always.  Dispose it before its backend, and only after every
/// [HipVectorBackend.searchAsync] that used it has completed.
final class HipRowMask {
  Pointer<Void> _mask;

  HipRowMask._(this._mask);

  /// Rows the mask keeps among the index's current node ids (tombstones not subtracted); -1 on failure.
  int get kept => _mask == nullptr ? -1 : HipVectorBackend._maskKept(_mask);

  void dispose() {
    if (_mask != nullptr) {
      HipVectorBackend._maskDestroy(_mask);
      _mask = nullptr;
    }
  }
}
