"""ctypes binding of libtostore_hip.so (include/tostore_hip.h).

The Python-side counterpart of the `dart:ffi` bridge in
tostore_amd/dart/tostore_hip_bridge.dart: same entry points, same status-code
convention (0 = ok, mirrors /root/reference/lib/src/handler/
system_ffi_helper.dart:219-262).  There is no fallback: a missing library or
a failing call raises.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (TSH_LIB_PATH: a development override -- tools/build_variants.py links probe variants of the library side by side)
LIB_PATH = os.environ.get("TSH_LIB_PATH") or os.path.join(_HERE, "libtostore_hip.so")

TSH_OK = 0
TSH_E_BAD_ARG = -1
TSH_E_DIM_MISMATCH = -2
TSH_E_OOM = -3
TSH_E_HIP = -4
TSH_E_NO_DEVICE = -5
TSH_E_OVERFLOW = -6
TSH_E_IO = -7
TSH_E_FORMAT = -8
TSH_E_BUSY = -9
TSH_E_RCCL = -10
TSH_E_PEER = -11
ABI_VERSION = 5

METRIC_L2, METRIC_IP, METRIC_COSINE = 0, 1, 2

c_i32, c_i64, c_f64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_double
p_f32 = ctypes.POINTER(ctypes.c_float)
p_f64 = ctypes.POINTER(ctypes.c_double)
p_i32 = ctypes.POINTER(ctypes.c_int32)
p_i64 = ctypes.POINTER(ctypes.c_int64)
p_u8 = ctypes.POINTER(ctypes.c_uint8)
p_void = ctypes.c_void_p


class TshCounters(ctypes.Structure):
    _fields_ = [
        ("rows", c_i64), ("deleted_rows", c_i64), ("searches", c_i64),
        ("scan_launches", c_i64), ("batch_launches", c_i64),
        ("fallback_searches", c_i64), ("candidates_total", c_i64),
        ("bytes_resident", c_i64), ("safe_mode", c_i32), ("device_id", c_i32),
        ("scan_us_sum", c_f64), ("scan_us_samples", c_i64),
        ("batch_kernel_last", c_i32), ("quarantined_rows", c_i32), ("exact_scans", c_i64),
        ("batch_plane_fallbacks", c_i64), ("batch_scan_fallbacks", c_i64), ("list_scans", c_i64),
        ("exact_redone", c_i64),
    ]


class TshNghInfo(ctypes.Structure):
    _fields_ = [
        ("dimensions", c_i32), ("metric", c_i32), ("precision", c_i32), ("page_size", c_i32),
        ("max_degree", c_i32), ("reserved", c_i32), ("next_node_id", c_i64), ("total_vectors", c_i64),
        ("deleted_count", c_i64), ("max_partition_file_size", c_i64), ("rows_loaded", c_i64),
        ("tombstones", c_i64), ("files_read", c_i64), ("pages_absent", c_i64), ("files_absent", c_i64),
        ("row_base", c_i64), ("row_end", c_i64),
    ]


class TshCommTimeline(ctypes.Structure):
    _fields_ = [
        ("calls", c_i64), ("queries", c_i64), ("groups", c_i64), ("retries", c_i64),
        ("world", c_i32), ("rank", c_i32), ("transport", c_i32), ("reserved", c_i32),
        ("call_us", c_f64), ("reserve_us", c_f64), ("wait_scan_us", c_f64), ("scan_us", c_f64),
        ("exchange_wait_us", c_f64), ("gather_us", c_f64), ("slice_d2h_us", c_f64), ("merge_us", c_f64),
        ("result_gather_us", c_f64), ("copy_out_us", c_f64), ("retry_scan_us", c_f64), ("pre_enqueue_us", c_f64),
    ]


# every symbol include/tostore_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "tsh_abi_version": (c_i32, []),
    "tsh_device_count": (c_i32, []),
    "tsh_last_error": (c_i32, [ctypes.c_char_p, c_i32]),
    "tsh_index_create": (c_i32, [c_i32, c_i32, c_i64, c_i32, ctypes.POINTER(p_void)]),
    "tsh_index_create_shard": (c_i32, [c_i32, c_i32, c_i64, c_i32, c_i64, ctypes.POINTER(p_void)]),
    "tsh_index_destroy": (c_i32, [p_void]),
    "tsh_index_append": (c_i32, [p_void, c_i64, c_i64, p_f32]),
    "tsh_index_append_device": (c_i32, [p_void, c_i64, c_i64, p_void]),
    "tsh_index_set_deleted": (c_i32, [p_void, p_i64, c_i64]),
    "tsh_index_load_rawvec_file": (c_i32, [p_void, ctypes.c_char_p, c_i32, c_i32, c_i64, c_i64, p_i64]),
    "tsh_index_open_ngh": (c_i32, [ctypes.c_char_p, c_i32, c_i32, ctypes.POINTER(p_void), ctypes.POINTER(TshNghInfo)]),
    "tsh_index_open_ngh_shard": (c_i32, [ctypes.c_char_p, c_i32, c_i32, c_i32, c_i32, ctypes.POINTER(p_void),
                                         ctypes.POINTER(TshNghInfo)]),
    "tsh_pq_train": (c_i32, [c_i32, p_f32, c_i64, c_i32, c_i32, c_i32, c_i32, ctypes.POINTER(ctypes.c_int32), p_f32]),
    "tsh_index_pq_encode": (c_i32, [p_void, c_i64, c_i64, p_f32, c_i32, c_i32, p_u8]),
    "tsh_index_size": (c_i64, [p_void]),
    "tsh_index_dim": (c_i32, [p_void]),
    "tsh_index_metric": (c_i32, [p_void]),
    "tsh_search": (c_i32, [p_void, p_f32, c_i32, c_i32, c_f64, p_u8, p_i64, p_f64, p_i32]),
    "tsh_mask_create": (c_i32, [p_void, p_u8, c_i64, ctypes.POINTER(p_void)]),
    "tsh_mask_destroy": (c_i32, [p_void]),
    "tsh_mask_kept": (c_i64, [p_void]),
    "tsh_search_masked": (c_i32, [p_void, p_f32, c_i32, c_i32, c_f64, p_void, p_i64, p_f64, p_i32]),
    "tsh_search_submit_masked": (c_i32, [p_void, p_f32, c_i32, p_void, p_i32]),
    "tsh_max_inflight": (c_i32, []),
    "tsh_search_submit": (c_i32, [p_void, p_f32, c_i32, p_u8, p_i32]),
    "tsh_search_ready": (c_i32, [p_void, c_i32]),
    "tsh_search_wait": (c_i32, [p_void, c_i32, c_f64, p_i64, p_f64, p_i32]),
    "tsh_candidate_block_bytes": (c_i64, [c_i32]),
    "tsh_default_block_entries": (c_i32, [c_i32]),
    "tsh_search_shard": (c_i32, [p_void, p_f32, c_i32, c_i32, p_u8, c_i32, p_void, p_void]),
    "tsh_search_shard_begin": (c_i32, [p_void, p_f32, c_i32, c_i32, p_u8, c_i32, p_void, c_i32, ctypes.POINTER(p_void)]),
    "tsh_search_shard_progress": (c_i32, [p_void, c_i32, p_i32]),
    "tsh_search_shard_end": (c_i32, [p_void]),
    "tsh_merge_candidates": (c_i32, [c_i32, c_i32, p_f32, c_i32, c_i32, c_f64, p_void, c_i32, c_i32,
                                     p_i64, p_f64, p_i32, p_i32]),
    "tsh_comm_unique_id": (c_i32, [p_void]),
    "tsh_comm_create": (c_i32, [p_void, c_i32, c_i32, c_i32, ctypes.POINTER(p_void)]),
    "tsh_comm_create_host": (c_i32, [c_i32, c_i32, c_i32, p_void, p_void, ctypes.POINTER(p_void)]),
    "tsh_comm_destroy": (c_i32, [p_void]),
    "tsh_comm_set_group": (c_i32, [p_void, c_i32]),
    "tsh_comm_world": (c_i32, [p_void]),
    "tsh_search_sharded": (c_i32, [p_void, p_void, p_f32, c_i32, c_i32, c_f64, p_u8, p_i64, p_f64, p_i32]),
    "tsh_comm_get_timeline": (c_i32, [p_void, ctypes.POINTER(TshCommTimeline), c_i32]),
    "tsh_get_counters": (c_i32, [p_void, ctypes.POINTER(TshCounters)]),
    "tsh_bench_scan": (c_i32, [p_void, p_f32, c_i32, p_u8, p_f64]),
    "tsh_bench_batch": (c_i32, [p_void, p_f32, c_i32, c_i32, c_i32, p_f64, p_f64]),
    "tsh_probe_scan_keys": (c_i32, [p_void, p_f32, p_f32, p_f32, p_f32]),
    "tsh_probe_batch_keys": (c_i32, [p_void, p_f32, c_i32, c_i32, p_f32, p_f32]),
    "tsh_probe_batch_row_band": (c_i32, [p_void, c_i32, p_f32, p_f32]),
    "tsh_index_set_option": (c_i32, [p_void, c_i32, c_i64]),
}


# tsh_allgather_fn: int32 (*)(void *user, const void *send, void *recv, int64 bytes)
ALLGATHER_FN = ctypes.CFUNCTYPE(c_i32, p_void, p_void, p_void, c_i64)


class TshError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libtostore_hip error {code}: {message}")
        self.code = code
        self.message = message


_lib = None


def lib() -> ctypes.CDLL:
    """Load the shared library (raises if it was not built: no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} is missing: build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950)")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.tsh_abi_version() != ABI_VERSION:
            raise RuntimeError("libtostore_hip.so ABI version mismatch")
        _lib = L
    return _lib


TSH_OPT_BATCH_MIN_NQ = 1
TSH_OPT_BATCH_KERNEL = 2
TSH_OPT_EXCHANGE_AHEAD = 3
TSH_OPT_EXACT_SCAN_ROWS = 4
TSH_OPT_EXACT_SELECT = 5
TSH_OPT_BATCH_HUB = 6
TSH_OPT_BATCH_GROUP = 7
TSH_OPT_TEST_HOOKS = 1000
TSH_TEST_HOOKS_MAGIC = 0x7465737468


def enable_test_hooks(on: bool = True) -> None:
    """Tests and rehearsals only: the library obeys TSH_RCCL_LIB / TSH_TEST_FAIL_ALLOC_OVER /
    TSH_SHARDS_SHARE_DEVICES only in a process that asked for it (include/tostore_hip.h, TSH_OPT_TEST_HOOKS)."""
    check(lib().tsh_index_set_option(None, TSH_OPT_TEST_HOOKS, TSH_TEST_HOOKS_MAGIC if on else 0))


def last_error() -> str:
    buf = ctypes.create_string_buffer(1024)
    lib().tsh_last_error(buf, 1024)
    return buf.value.decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc != TSH_OK:
        raise TshError(rc, last_error())
