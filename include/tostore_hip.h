/*
 * tostore_hip.h -- C-ABI of libtostore_hip.so: MI355X (gfx950) exhaustive kNN
 * behind ToStore's vectorSearch().
 *
 * The reference (tocreator/tostore v3.2.0, pure Dart) has no plugin/FFI hook
 * for vector search; the seam this library plugs into is the single call
 *   NghGraphEngine.search(...)   lib/src/core/ngh_graph_engine.dart:67-135
 * made from VectorIndexManager.vectorSearch
 *                                lib/src/core/vector_index_manager.dart:538-548
 * plus the data-feed seams listed per function below.  Conventions mirror the
 * reference's only existing FFI (lib/src/handler/system_ffi_helper.dart:21-55,
 * 219-262): int32 status returns with 0 = success, caller-allocated
 * out-params, no callbacks, every failure recoverable by falling back to the
 * Dart path.  The `dart:ffi` binding a maintainer would add is in
 * INTEGRATION.md and tostore_amd/dart/tostore_hip_bridge.dart.
 *
 * Ownership: the library owns only tsh_index handles and device memory.
 * Every host pointer is caller-allocated and caller-freed; inputs are fully
 * consumed before the call returns; no pointer is retained.
 * Threading: every entry point is thread-safe; searches on one handle may
 * run concurrently, append/delete/load take the handle exclusively.
 * Errors: <0 = error class below; text via tsh_last_error (thread-local).
 * The library never aborts, throws across the boundary, or falls back to a
 * CPU implementation: without a usable GPU every compute entry returns
 * TSH_E_NO_DEVICE.
 */
#ifndef TOSTORE_HIP_H
#define TOSTORE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: tsh_counters grew (fused_launches) and tsh_ngh_info grew (pages_absent, files_absent) after version 1 shipped; a
 * host built against the version-1 structs would be written past its buffers, so the number changed with them.
 * tsh_comm_create_host / tsh_comm_set_group and TSH_E_PEER came with version 2 as well. */
#define TSH_ABI_VERSION 5
/* 3: tsh_counters grew again (batch_plane_fallbacks, batch_scan_fallbacks); tsh_comm_get_timeline / tsh_comm_timeline
 * came with it.
 * 4: tsh_search_shard_begin / _progress / _end (the progressive shard search tsh_search_sharded is now built on);
 *    tsh_counters.list_scans, tsh_counters.exact_scans (the reserved slot of version 2), tsh_comm_timeline.pre_enqueue_us;
 *    TSH_OPT_EXCHANGE_AHEAD, TSH_OPT_EXACT_SCAN_ROWS, TSH_OPT_TEST_HOOKS.
 * 5: mask handles (tsh_mask_create / _destroy / _kept, tsh_search_masked, tsh_search_submit_masked): a WHERE row set
 *    that lives on the device across queries; tsh_index_open_ngh_shard (a rank cold-starts its own row range);
 *    tsh_ngh_info grew (row_base, row_end); tsh_comm_timeline's sampled fields are scaled by exchanges / timed
 *    exchanges instead of a constant. */

/* status codes */
#define TSH_OK 0
#define TSH_E_BAD_ARG (-1)
#define TSH_E_DIM_MISMATCH (-2)
#define TSH_E_OOM (-3)
#define TSH_E_HIP (-4)
#define TSH_E_NO_DEVICE (-5)
#define TSH_E_OVERFLOW (-6) /* a candidate block was too small; retry with more entries */
#define TSH_E_IO (-7)
#define TSH_E_FORMAT (-8)
#define TSH_E_BUSY (-9) /* tsh_search_submit: tsh_max_inflight() asynchronous searches are already un-waited on the
                           handle, or an append / delete is waiting for the open ones: wait for them, then submit again */
#define TSH_E_RCCL (-10) /* librccl could not be loaded, or a collective failed */
#define TSH_E_PEER (-11) /* tsh_search_sharded: another rank of the collective failed (its own call returned the real
                            error); nothing was written on this rank.  The communicator stays usable. */

/* metric = enum order of VectorDistanceMetric, lib/src/model/table_schema.dart:2511-2531 */
#define TSH_METRIC_L2 0
#define TSH_METRIC_IP 1
#define TSH_METRIC_COSINE 2

typedef struct tsh_index tsh_index;

/* counters: replaces nothing in the reference (it has no tracing on this
 * path); feeds Logger-style diagnostics on the Dart side */
typedef struct tsh_counters {
  int64_t rows;              /* rows ever appended (= next row id) */
  int64_t deleted_rows;      /* tombstoned rows */
  int64_t searches;          /* queries answered */
  int64_t scan_launches;     /* single-query scan kernels launched */
  int64_t batch_launches;    /* batched (MFMA) passes launched */
  int64_t fallback_searches; /* queries that took the wide-band fallback */
  int64_t candidates_total;  /* rows re-ranked in f64 */
  int64_t bytes_resident;    /* device bytes held by this handle */
  int32_t safe_mode;         /* !=0: more than 1024 rows of one shard hold values outside the f32 error model
                                (see quarantined_rows); every search re-ranks all rows in f64 (exact, slow) */
  int32_t device_id;
  /* scan-kernel device time sampled with HIP events on the stream the kernel runs
   * on, inside real searches (every 4th query): sum of microseconds / samples */
  double scan_us_sum;
  int64_t scan_us_samples;
  int32_t batch_kernel_last; /* TSH_OPT_BATCH_KERNEL variant the last batched search ran (0/1/2; -1 none yet) */
  int32_t quarantined_rows;  /* live rows outside the f32 error model (non-finite or > 1e15 elements; cosine: norm
                                below 2^-50) that are kept out of the scan and re-ranked exactly on every search */
  int64_t exact_scans;       /* of scan_launches: searches with at most TSH_OPT_EXACT_SCAN_ROWS rows to look at (a
                                selective mask's list, a small index or shard), answered by the exact f64 sums of all of
                                them and a select among the exact distances -- two dispatches, no f32 keys, no band.
                                (The slot was "fused_launches", reserved and 0, before ABI 4.) */
  /* A batched call whose device allocations fail degrades instead of failing (results are identical on every path): */
  int64_t batch_plane_fallbacks; /* calls that could not allocate the fp16 / bf16x3 copy of the rows and scored
                                    their queries with the f32 MFMA kernel on the rows as stored (no copy needed) */
  int64_t batch_scan_fallbacks;  /* calls whose batch scratch could not be allocated either and that were answered by
                                    pipelined single-query scans */
  int64_t list_scans;            /* of scan_launches: scans of a selective row mask as a compacted list of row ids
                                    (scan_list_kernel) instead of a walk over the tiles */
  int64_t exact_redone;          /* of exact_scans: searches whose wide pick (TSH_OPT_EXACT_SELECT) met a cut bin too full
                                    for its block -- ties by the hundred, a k-th neighbour outside the key histogram's
                                    window -- and were finished by the one-workgroup select on the same keys (ABI 5) */
} tsh_counters;

int32_t tsh_abi_version(void);
/* number of usable HIP devices; 0 when none (never an error) */
int32_t tsh_device_count(void);
/* copies the calling thread's last error text (NUL-terminated, truncated to
 * len); returns the untruncated length */
int32_t tsh_last_error(char *buf, int32_t len);

/* Create an empty device index on the calling thread's current HIP device
 * (n_devices == 1) or row-range sharded over devices 0..n_devices-1.
 * Created lazily by the Dart side on the first vectorSearch / writeChanges of
 * an index (the reference loads nothing vector-related at open:
 * lib/src/core/data_store_impl.dart:822-846, vector_index_manager.dart:39-43).
 * dim/metric come from NghIndexMeta (lib/src/model/ngh_index_meta.dart:82-100).
 * 1 <= dim <= 4096 (a 16 KiB raw-vector page holds float32 vectors up to
 * dim 4073, lib/src/core/ngh_page.dart:575-579). */
int32_t tsh_index_create(int32_t dim, int32_t metric, int64_t capacity_rows,
                         int32_t n_devices, tsh_index **out);

/* One shard of a row-range partitioned corpus, for one-process-per-GPU
 * deployments: holds global rows [row_base, row_base + size) on device
 * `device_id` (-1 = current).  Ids reported by this handle are global. */
int32_t tsh_index_create_shard(int32_t dim, int32_t metric, int64_t capacity_rows,
                               int32_t device_id, int64_t row_base, tsh_index **out);

/* Replaces cache teardown: VectorIndexManager.clearCacheForTable/Index, dispose
 * (vector_index_manager.dart:1192-1216); also required after reorderByLocality
 * renumbers node ids (:932-1159). */
int32_t tsh_index_destroy(tsh_index *idx);

/* Append (or overwrite) rows [first_row_id, first_row_id + n_rows); `rows` is
 * row-major n_rows x dim float32, already converted exactly as the reference
 * stores them (f64 -> f32, truncate / zero-pad: core/compute/
 * vector_batch_prepare_compute.dart:79-86).  Mirrors NghGraphEngine.
 * _writeRawVector / insertBatch (ngh_graph_engine.dart:691-720, :297-403):
 * node ids are dense and monotonically assigned (:321).  Ids skipped by a gap
 * are treated as absent rows (never returned), like a missing raw-vector page
 * (ngh_graph_engine.dart:124-125).  Shard handles take GLOBAL ids. */
int32_t tsh_index_append(tsh_index *idx, int64_t first_row_id, int64_t n_rows,
                         const float *rows);
/* Same, from a device pointer on the handle's device (bulk loaders that
 * already hold the column in HBM).  The copy runs on a library stream: the
 * caller must have synchronised whatever produced d_rows before the call. */
int32_t tsh_index_append_device(tsh_index *idx, int64_t first_row_id, int64_t n_rows,
                                const void *d_rows);

/* Tombstones: mirrors NghGraphEngine.deleteBatch (ngh_graph_engine.dart:411-445),
 * which sets NghNodeFlags.deleted (ngh_page.dart:105-108) and leaves the raw
 * vector bytes in place.  Unlike the reference's beam search (which can leak
 * deleted rows from other pages, :230-232) deleted rows are never returned.
 * Unknown ids are ignored. */
int32_t tsh_index_set_deleted(tsh_index *idx, const int64_t *ids, int64_t n);

/* Cold load of one raw-vector partition file
 * <index>/ngh/rawvec/dir_N/pP.ngh (lib/src/core/path_manager.dart:318-324):
 * data pages 1..last of `page_size` bytes (page 0 is the partition meta page,
 * core/ngh_page.dart:29-98), each a 20-byte 'TPG2' header + CRC32
 * (core/btree_page.dart:132-234) + NghRawVectorPage payload
 * (core/ngh_page.dart:310-450; f64 / f32 / i8 elements converted to f32 exactly
 * as getVectorAsFloat32 does, :364-391).  `precision` is meta.json's
 * VectorPrecision index (0 f64, 1 f32, 2 i8) and fixes vectorsPerRawPage
 * (core/ngh_page.dart:575-579); data page p holds node ids
 * first_row_id + (p-1)*vectorsPerRawPage + slot (model/ngh_index_meta.dart:480-490).
 * Reader semantics follow readRawVectorPage (core/ngh_partition_manager.dart:239-295)
 * with ONE deliberate difference: where the reference makes up a page of zero
 * vectors (a page past the end of the file, :270-281) the ids stay ABSENT rows here --
 * the reference only meets such a slot if its graph walk reaches it, an exhaustive
 * scan would return those zero rows to every query.  A bad magic / type / CRC is an
 * error (the reference throws: core/btree_page.dart:215-233), and so is a page that
 * passes its CRC but does not decode as a raw-vector payload (ciphertext of an index
 * written with encryptVectorIndex): TSH_E_FORMAT, nothing appended from that page on.
 * Only ids < first_row_id + max_rows are loaded (meta.nextNodeId bounds them).
 * *out_rows = rows appended. */
int32_t tsh_index_load_rawvec_file(tsh_index *idx, const char *path, int32_t page_size,
                                   int32_t precision, int64_t first_row_id, int64_t max_rows,
                                   int64_t *out_rows);

/* What tsh_index_open_ngh found: the NghIndexMeta fields it used
 * (lib/src/model/ngh_index_meta.dart:359-408) and what it loaded. */
typedef struct tsh_ngh_info {
  int32_t dimensions;
  int32_t metric;    /* 0 l2, 1 innerProduct, 2 cosine */
  int32_t precision; /* 0 float64, 1 float32, 2 int8 (VectorPrecision order) */
  int32_t page_size;
  int32_t max_degree;
  int32_t reserved;
  int64_t next_node_id;
  int64_t total_vectors; /* meta.json's own count of live vectors */
  int64_t deleted_count; /* meta.json's own count of tombstones */
  int64_t max_partition_file_size;
  int64_t rows_loaded; /* node ids now resident */
  int64_t tombstones;  /* node ids whose graph slot carries NghNodeFlags.deleted */
  int64_t files_read;
  int64_t pages_absent; /* raw-vector pages below nextNodeId that are not on disk (file missing or shorter): their
                           node ids are ABSENT rows; a healthy index has 0 -- a caller may refuse the handle otherwise */
  int64_t files_absent; /* raw-vector partition files below nextNodeId that are missing altogether */
  int64_t row_base;     /* node ids [row_base, row_end) are what this handle was opened for: 0 and nextNodeId for */
  int64_t row_end;      /* tsh_index_open_ngh, the rank's range for tsh_index_open_ngh_shard (ABI 5) */
} tsh_ngh_info;

/* Cold start from an index directory written by the reference, without Dart
 * (SURVEY.md section 8f, N1): `<index>/ngh` as laid out by
 * lib/src/core/path_manager.dart:275-324.  Reads meta.json (jsonEncode of
 * NghIndexMeta.toJson, vector_index_manager.dart:623-634) for dimensions,
 * metric, precision, nextNodeId, page and partition-file sizes; creates the
 * handle; loads every raw-vector partition rawvec/dir_{p / max_entries_per_dir}/
 * p{p}.ngh with the addressing of ngh_index_meta.dart:480-490; then walks the
 * graph partitions and tombstones every node whose slot flags carry
 * NghNodeFlags.deleted (ngh_page.dart:105-108,198-213).  A missing graph file /
 * page reads as "no flags" exactly as in the reference; a missing raw-vector
 * file / page leaves its node ids ABSENT (never returned; counted in
 * info->pages_absent / files_absent -- see tsh_index_load_rawvec_file); a page
 * with a bad frame, type or CRC is TSH_E_FORMAT (the reference throws there,
 * btree_page.dart:215-233).  max_entries_per_dir <= 0 selects the reference
 * default of 500 (handler/common.dart:43).  Encrypted vector pages are not
 * supported (EncryptionConfig.encryptVectorIndex must be off): their payloads
 * pass the CRC but do not decode -> TSH_E_FORMAT.  info may be NULL. */
int32_t tsh_index_open_ngh(const char *ngh_dir, int32_t max_entries_per_dir, int32_t n_devices, tsh_index **out,
                           tsh_ngh_info *info);
/* Cold start of ONE row-range shard, for the one-process-per-GPU deployment tsh_search_sharded serves: rank `rank`
 * of `world` opens node ids [rank * per, min(nextNodeId, (rank + 1) * per)), per = ceil(nextNodeId / world), on
 * `device` (-1 = current) as a shard handle (tsh_index_create_shard: ids stay GLOBAL).  The addressing of
 * lib/src/model/ngh_index_meta.dart:480-490 makes that range a contiguous run of raw-vector partition files
 * (lib/src/core/path_manager.dart:275-324) and, inside the first and the last of them, of pages: only those files
 * are opened and only those pages read -- at BASELINE.json's C4 (10 M x 1536: 82 GB of pages) a rank of eight
 * reads its eighth.  A range that starts or ends inside a page takes that page's slots from / up to the boundary.
 * Tombstones come from the graph pages that hold the range's node ids (lib/src/core/ngh_page.dart:105-108,
 * 198-213), likewise only those.  Everything else -- meta.json, page checks, absent pages and files, errors -- is
 * tsh_index_open_ngh's; info (nullable) reports the census of THIS range (rows_loaded, tombstones, files_read,
 * pages_absent, files_absent) and the range itself (row_base, row_end).  W shards opened this way and merged
 * (tsh_merge_candidates / tsh_search_sharded) answer exactly what the whole-index handle answers. */
int32_t tsh_index_open_ngh_shard(const char *ngh_dir, int32_t max_entries_per_dir, int32_t device, int32_t world,
                                 int32_t rank, tsh_index **out, tsh_ngh_info *info);

/* Write-path helper: PQ-encode resident rows
 * [first_row_id, first_row_id + n_rows) against a trained codebook -- the
 * arithmetic of batchPqEncode (lib/src/core/compute_tasks.dart:2292-2326) ==
 * VectorQuantizer.encode (core/vector_quantizer.dart:357-368,461-483): per
 * sub-space the FIRST centroid with the smallest f64-accumulated squared
 * distance.  codebook: host, float32, layout centroids[(m*K + k)*subDim + d]
 * (vector_quantizer.dart:15), subDim = dim / subspaces; out_codes: host,
 * n_rows x subspaces bytes (one NghPqCodePage entry each, ngh_page.dart:232-300).
 * Codes are bit-exact with the reference. */
int32_t tsh_index_pq_encode(tsh_index *idx, int64_t first_row_id, int64_t n_rows, const float *codebook,
                            int32_t subspaces, int32_t centroids, uint8_t *out_codes);

/* Write-path helper (N4): train a PQ codebook the way the index manager's isolate
 * tasks do -- one trainPqSubspace per sub-space (lib/src/core/compute_tasks.dart:
 * 2135-2266, dispatched from core/vector_index_manager.dart:740-850 when >= 100
 * samples were collected; `iterations` is 10 there and centroids = min(256, n)).
 * samples: host, n x dim float32 (already _toFloat32'd, cosine rows normalised);
 * init_index: host, subspaces x centroids sample indices -- the values the
 * reference draws with Random(42 + subspaceIndex).nextInt(n), which stay on the
 * Dart side; out_codebook: host, layout centroids[(m*K + k)*subDim + d]
 * (vector_quantizer.dart:15).  Given the same init_index the codebook is
 * bit-identical to the reference's (mixed f32/f64 arithmetic restated, see
 * tsh_pq.hip.h).  Stateless: runs on `device`, needs no index handle. */
int32_t tsh_pq_train(int32_t device, const float *samples, int64_t n, int32_t dim, int32_t subspaces,
                     int32_t centroids, int32_t iterations, const int32_t *init_index, float *out_codebook);

int64_t tsh_index_size(tsh_index *idx); /* next row id (rows incl. tombstones) */
int32_t tsh_index_dim(tsh_index *idx);
int32_t tsh_index_metric(tsh_index *idx);

/* The drop-in for NghGraphEngine.search (ngh_graph_engine.dart:67-135) with
 * ef -> infinity: exact distances of EVERY live row, reference phase-3
 * semantics (:122-134): distance = sqrt(sum) for L2, -dot for IP,
 * 1 - cos for cosine, all in f64 accumulated left to right over f32 elements
 * (:908-946); rows with distance > distance_threshold dropped (NaN = none);
 * ascending by Dart double.compareTo, ties by row id; at most k per query.
 *   queries     nq x dim float32, prepared by the caller exactly as
 *               vector_index_manager.dart:514-520 does (_toFloat32, and
 *               _normalizeFloat32 for cosine)
 *   row_mask    nullable; ceil(size/8) bytes, bit i (LSB first) = 1 keeps row
 *               i (WHERE pre-filter; new capability).  Bytes read
 *               scale with the rows kept: a mask that keeps fewer than one row
 *               in 24 is compacted into a list of row ids once per call and
 *               scanned as a gather; a one-range mask (WHERE id BETWEEN ...)
 *               costs a batched call no more than a scattered one
 *   out_ids     nq x k      out_dist nq x k      out_count nq
 * nq == 1 (or small) streams the corpus once per query (HBM-bound kernel);
 * larger nq uses the batched matrix-core path, whose last steps (exact f64
 * re-rank, distance, threshold, order, cut) run on the device.  A device too
 * full for that path's buffers degrades the call (f32 matrix-core kernel on the
 * rows as stored, then pipelined scans: tsh_counters.batch_*_fallbacks) instead
 * of failing it.  Thread-safe;
 * two batched calls on one handle overlap (one prepares / copies out while the
 * GPU works on the other), further concurrent ones queue.  Empty index or
 * k <= 0 returns TSH_OK with counts 0 (the reference returns const []: :78). */
int32_t tsh_search(tsh_index *idx, const float *queries, int32_t nq, int32_t k,
                   double distance_threshold, const uint8_t *row_mask,
                   int64_t *out_ids, double *out_dist, int32_t *out_count);

/* ---- mask handles: a WHERE row set that lives on the device ------------------------------------------------------
 * tsh_search's row_mask pointer is consumed per call: the library slices the caller's bitmap, counts it and -- for a
 * selective one -- lists its rows, on the host, every time (three passes over 125 KB at 1 M rows: more than half of
 * a lone masked query's latency).  A row set that serves many queries -- the rows a WHERE clause keeps, mapped from
 * primary keys through the pk -> nodeId tree (lib/src/core/vector_index_manager.dart:1223-1378), or the complement
 * of a tombstone set (lib/src/core/ngh_page.dart:105-108) -- is made a HANDLE instead: the bitmap is uploaded once,
 * and when it is selective (few enough kept rows for the list scan or the exact path) compacted into the ascending
 * list of kept row ids ON THE DEVICE (popcount per 64-row word, a prefix over the words, one thread per word writing
 * its rows: the list never exists on the host).  Searches with the handle read both in place: no per-call copy, no
 * host pass.
 *   bits     GLOBAL keep bitmap, bit i (LSB first) = 1 keeps row id i, n_bytes bytes of it; rows at or beyond
 *            8 * n_bytes are NOT kept -- rows appended after the mask was made included (the handle re-slices its
 *            copy of the bitmap on the first search after an append; never the caller's memory again)
 *   rows tombstoned later are dropped as always (the kernels check the live bitmap), deleted rows never come back
 * A handle belongs to the index it was made for (any handle kind: whole index, several devices, a shard) and must be
 * destroyed before it; destroy it only after every ticket submitted with it has been waited for.  Thread-safe: any
 * number of searches may share one handle.  Results are identical to the pointer form's. */
typedef struct tsh_mask tsh_mask;
int32_t tsh_mask_create(tsh_index *idx, const uint8_t *bits, int64_t n_bytes, tsh_mask **out);
int32_t tsh_mask_destroy(tsh_mask *mask);
/* rows the mask keeps among the index's current row ids (tombstones not subtracted); < 0 = error */
int64_t tsh_mask_kept(tsh_mask *mask);
/* tsh_search / tsh_search_submit with the row set of a handle (mask may be NULL: no filter) */
int32_t tsh_search_masked(tsh_index *idx, const float *queries, int32_t nq, int32_t k, double distance_threshold,
                          tsh_mask *mask, int64_t *out_ids, double *out_dist, int32_t *out_count);
int32_t tsh_search_submit_masked(tsh_index *idx, const float *query, int32_t k, tsh_mask *mask, int32_t *out_ticket);

/* Asynchronous form of a single-query tsh_search, for callers that keep several
 * independent queries in flight (the reference serves concurrent vectorSearch
 * calls from its isolate pool, lib/src/Interface/compute_native.dart:18-300): a
 * submitted query's select / re-rank / copies overlap the next query's corpus
 * scan.  At most tsh_max_inflight() un-waited tickets per handle (TSH_E_BUSY
 * beyond).  The handle is share-locked from submit to wait: appends and deletes
 * block until every ticket is waited.  Every ticket must be waited exactly once
 * before tsh_index_destroy.  Results are identical to tsh_search. */
int32_t tsh_max_inflight(void);
int32_t tsh_search_submit(tsh_index *idx, const float *query, int32_t k, const uint8_t *row_mask,
                          int32_t *out_ticket);
/* 1 = the ticket's kernels have finished (tsh_search_wait will not wait for the GPU), 0 = still running, < 0 =
 * error.  Never blocks: a single-threaded host (a Dart isolate, whose yield budget is 8 ms on clients and 50 ms on
 * servers: lib/src/model/data_store_config.dart:225-230, core/yield_controller.dart:110-169) submits, returns to
 * its event loop, and waits once this says 1. */
int32_t tsh_search_ready(tsh_index *idx, int32_t ticket);
int32_t tsh_search_wait(tsh_index *idx, int32_t ticket, double distance_threshold, int64_t *out_ids,
                        double *out_dist, int32_t *out_count);

/* ---- row-sharded deployments (one process per GPU) ----------------------
 * Each rank scans its shard and emits, per query, one fixed-size candidate
 * block into DEVICE memory; ranks all-gather the blocks (RCCL) and any rank
 * merges them on the host.  A block = 64-byte header + entries x 24 bytes
 * {int64 global id, f64 sum0, f64 sum1} holding the exact f64 accumulations
 * of ngh_graph_engine.dart:920-946 for every row that can be in the shard's
 * top k. */
int64_t tsh_candidate_block_bytes(int32_t entries);
int32_t tsh_default_block_entries(int32_t k);
/* d_out_blocks: device buffer of nq * tsh_candidate_block_bytes(entries) on the
 * handle's device; the kernels store the blocks into it directly and the call
 * returns after they completed (host-synchronised), so any stream may read it
 * next.  stream: reserved (pass NULL or the consumer's hipStream_t) -- a block is
 * final only once the HOST has seen its header (ties that overflow a block's
 * candidate list are redone by a wider pass the host starts), so completion
 * cannot be handed to a stream; a caller that wants to work on the first blocks
 * while the later ones are still being scanned uses the progressive form below.
 * row_mask is GLOBAL (bit = global row id).  Rows kept out of the scan (tsh_counters.quarantined_rows) are
 * appended to every block the mask lets them into; a block they do not fit in reports count > entries like any
 * other overflow, and tsh_merge_candidates answers TSH_E_OVERFLOW with the entry count to retry with. */
int32_t tsh_search_shard(tsh_index *idx, const float *queries, int32_t nq, int32_t k,
                         const uint8_t *row_mask, int32_t entries, void *d_out_blocks,
                         void *stream);
/* Progressive form of tsh_search_shard, for callers that exchange the blocks in
 * groups (tsh_search_sharded is built on it; tostore_amd/sharded.py can be): the
 * scans of ALL nq queries run as one pipeline on a library thread -- the GPU sees
 * no group boundaries, where group-by-group tsh_search_shard calls pay the fill and
 * drain of the scan pipeline per call (~45 us on a 125 k x 768 shard) -- and the
 * blocks become final in query order.
 *   _begin     starts the search and returns at once; queries / row_mask are copied
 *              (no pointer is retained), d_out_blocks must stay valid until _end.
 *              step: the caller's group size (queries it will ask for at a time);
 *              where a call of `step` queries would go to the matrix cores the
 *              search proceeds in calls of `step` queries, otherwise every query is
 *              its own scan.  0 = nq.
 *   _progress  blocks until the first min(want, nq) queries' blocks are final in
 *              d_out_blocks (host-synchronised: any stream may read them next) and
 *              returns TSH_OK, or until the search failed before it got there and
 *              returns its error; *out_done (nullable) = leading queries final.
 *   _end       waits for whatever still runs, frees the handle, returns the
 *              search's status.  Must be called exactly once per _begin.
 * Blocks, masks, overflow protocol: exactly tsh_search_shard's.  No reference
 * counterpart (the reference has no distributed compute). */
typedef struct tsh_shard_stream tsh_shard_stream;
int32_t tsh_search_shard_begin(tsh_index *idx, const float *queries, int32_t nq, int32_t k,
                               const uint8_t *row_mask, int32_t entries, void *d_out_blocks,
                               int32_t step, tsh_shard_stream **out);
int32_t tsh_search_shard_progress(tsh_shard_stream *stream, int32_t want, int32_t *out_done);
int32_t tsh_search_shard_end(tsh_shard_stream *stream);
/* Host-side merge of n_blocks x nq candidate blocks (layout [block][query]),
 * applying the final sqrt / negate / 1-cos, threshold, ordering and top-k cut.
 * Pure host code (no device needed).  Returns TSH_E_OVERFLOW if any block was
 * truncated: every rank sees the same headers, so all ranks retry with the
 * entry count written to *needed_entries. */
int32_t tsh_merge_candidates(int32_t metric, int32_t dim, const float *queries,
                             int32_t nq, int32_t k, double distance_threshold,
                             const void *blocks, int32_t n_blocks, int32_t entries,
                             int64_t *out_ids, double *out_dist, int32_t *out_count,
                             int32_t *needed_entries);

/* ---- the same exchange from a host without torch (a Dart process per GPU) ------------------------------------
 * RCCL all-gather of the per-shard candidate blocks over xGMI + host merge, behind plain C.  librccl is loaded
 * with dlopen on first use.  One communicator per rank:
 *   rank 0:     tsh_comm_unique_id(id)  -> ship the 128 bytes to the other ranks (any channel the host has)
 *   every rank: tsh_comm_create(id, world, rank, device, &comm)      (collective: returns when all ranks called)
 *   every rank: tsh_search_sharded(shard, comm, queries, ...)        (collective: same queries / nq / k / threshold
 *               on every rank; identical answer on every rank -- the one tsh_search gives on one un-sharded
 *               index over the same rows)
 *   every rank: tsh_comm_destroy(comm)
 * Inside one call the queries travel in groups: this rank's scans of ALL the call's queries run as one pipeline
 * (tsh_search_shard_begin), and every group is all-gathered (device to device) and merged as soon as its blocks
 * are final, while the scans of the later groups go on.  Of every group each
 * rank copies back and merges only its own slice of the queries (world blocks per query); a second, small
 * all-gather hands every slice's final ids / distances to every rank.  (Groups of at most 128 blocks in all -- world x
 * queries -- are merged whole on every rank instead, and the second all-gather does not happen.)
 * Failures: a rank whose own part fails (bad handle, TSH_E_BUSY, a HIP error in its shard search, an allocation)
 * STAYS in the collective, contributes blocks that say so, and returns its error; the other ranks return
 * TSH_E_PEER; the communicator remains usable.  Only a failing all-gather / stream operation (TSH_E_RCCL,
 * TSH_E_HIP from the exchange itself) leaves the ranks out of step: destroy the communicator then.
 * row_mask is GLOBAL.  No reference counterpart (the reference has no distributed compute). */
typedef struct tsh_comm tsh_comm;
#define TSH_COMM_ID_BYTES 128
int32_t tsh_comm_unique_id(void *out_id);
int32_t tsh_comm_create(const void *id, int32_t world, int32_t rank, int32_t device, tsh_comm **out);
/* The same protocol over a transport the host brings (ranks on several nodes, a host without RCCL, tests with
 * several ranks on one GPU): `allgather(user, send, recv, bytes)` must place every rank's `bytes` at
 * recv + rank * bytes on every rank (host memory, synchronous; 0 = ok) -- the one callback in this ABI; it is
 * invoked on the thread that called tsh_search_sharded, never after that call returned.  The candidate blocks
 * then travel device -> host -> callback instead of device -> device. */
typedef int32_t (*tsh_allgather_fn)(void *user, const void *send, void *recv, int64_t bytes);
int32_t tsh_comm_create_host(int32_t world, int32_t rank, int32_t device, tsh_allgather_fn allgather, void *user,
                             tsh_comm **out);
int32_t tsh_comm_destroy(tsh_comm *comm);
int32_t tsh_comm_world(tsh_comm *comm);
/* queries per exchange of tsh_search_sharded; 0 (default) = the library's own schedule.  Calls of up to 128 queries that
 * the shards scan query by query (no rank batches -- TSH_OPT_BATCH_MIN_NQ = 0 on every handle -- or the cost model says
 * scans) go in SHRINKING groups (half of what is left each time, never below what hides an exchange behind the scans that
 * follow -- sized by the bytes a scan of the largest shard reads -- nor below four: 20 queries on 125 k x 768 shards as
 * 10 + 5 + 5), because only the LAST group's exchange is exposed; calls the shards answer on their matrix cores go as ONE
 * group (a group is one batched call per shard).  Beyond 128 queries: 256 per group, 512 from 1024 queries on.  What the
 * ranks know of each other (shard sizes, whether they batch) they tell each other whenever their buffers grow and on every
 * 64th call: a changed option takes that long to reach the schedule.  n > 0: uniform groups of n.  Same value on every rank. */
int32_t tsh_comm_set_group(tsh_comm *comm, int32_t queries_per_exchange);
int32_t tsh_search_sharded(tsh_index *shard, tsh_comm *comm, const float *queries, int32_t nq, int32_t k,
                           double distance_threshold, const uint8_t *row_mask, int64_t *out_ids, double *out_dist,
                           int32_t *out_count);

/* Where the time of this rank's tsh_search_sharded calls went: sums over the calls since the communicator was made
 * (or since the last reset).  On the calling thread a call is, group after group,
 *   reserve | pre_enqueue | wait_scan | exchange_wait (= all-gather of the blocks + this rank's slice to the host) |
 *   merge | result_gather | copy_out        (+ retry_scan for groups redone with larger blocks)
 * so these eight add up to call_us but for loop overhead; scan_us runs beside them on the helper thread (group g + 1
 * is scanned while group g is exchanged), and gather_us + slice_d2h_us are the device-side split of exchange_wait_us.
 * A rank that waits for slower peers shows it in gather_us (the all-gather cannot finish before the last rank
 * joins); a rank whose own scans are the bottleneck shows it in wait_scan_us.  No reference counterpart. */
typedef struct tsh_comm_timeline {
  int64_t calls;   /* tsh_search_sharded calls that reached their exchange */
  int64_t queries; /* queries of those calls */
  int64_t groups;  /* exchanges (retried ones count again) */
  int64_t retries; /* groups redone with larger candidate blocks */
  int32_t world;
  int32_t rank;
  int32_t transport; /* 0 RCCL, 1 host callback, 2 a library named by TSH_RCCL_LIB (tests) */
  int32_t reserved;
  double call_us;          /* wall time inside the calls */
  double reserve_us;       /* buffers (+ the agreement all-gather when they grew) */
  double wait_scan_us;     /* calling thread blocked until this rank's scans of the group were done */
  double scan_us;          /* helper thread: tsh_search_shard of the groups (overlaps the previous group's exchange) */
  double exchange_wait_us; /* from issuing the block all-gather until this rank's slice is on the host */
  double gather_us;        /* the all-gather (RCCL: device time on the communicator's stream; every fourth exchange is
                              timed, the sum scaled by exchanges / timed exchanges) */
  double slice_d2h_us;     /* the slice's copy to the host (sampled likewise) */
  double merge_us;         /* host merge of the slice (threshold, order, cut) */
  double result_gather_us; /* every slice's results to every rank: H2D + all-gather + D2H */
  double copy_out_us;      /* into the caller's arrays */
  double retry_scan_us;    /* re-scans of overflowing groups, on the calling thread */
  double pre_enqueue_us;   /* RCCL transport: launching a group's all-gather + copy AHEAD of its blocks, behind a gate
                              the host opens when they are final (the launch's host cost hides behind the scans) */
} tsh_comm_timeline;
int32_t tsh_comm_get_timeline(tsh_comm *comm, tsh_comm_timeline *out, int32_t reset);

int32_t tsh_get_counters(tsh_index *idx, tsh_counters *out);

/* ---- measurement hooks (bench.py / profiles) -----------------------------
 * Average device time in microseconds of the scan kernel alone, measured with
 * HIP events on the library's own stream over `iters` launches of one query. */
int32_t tsh_bench_scan(tsh_index *idx, const float *query, int32_t iters,
                       const uint8_t *row_mask, double *out_avg_us);
/* One real batched search of nq queries, `iters` times: average device time of
 * the two matrix-core passes (sample + filtered) that together score every row
 * once, in microseconds (HIP events on the library's stream), and the
 * algorithmic flop count 2*nq*rows*dim they performed. */
int32_t tsh_bench_batch(tsh_index *idx, const float *queries, int32_t nq, int32_t k, int32_t iters,
                        double *out_avg_gemm_us, double *out_flops);

/* Probes of the pre-filter keys (tests only: tests/test_gpu_bands.py holds every key kernel to the error bound the
 * host claims for it, on rows built to err in one direction).  Single-shard handles.
 * tsh_probe_scan_keys: the single-query scan kernel's f32 ranking key of every row (NaN = row not live) and the
 * band the select step would apply: a row of the true top k has key <= tau * (1 + eps_rel) + delta_abs, where
 * eps_rel = 3 eps and delta_abs = 2 delta * 1.0001 for the per-key bounds |key - exact| <= eps * exact (L2) resp.
 * <= delta (inner product, cosine); exact = sum (q - v)^2, -q.v, -q.v / |v|.
 * tsh_probe_batch_keys: the batched key kernel's (TSH_OPT_BATCH_KERNEL) keys of every row for nq queries, row-major
 * nq x size, and per query 2 * 1.0001 * (the bound on |key - exact|); L2 keys are |q|^2 + |v|^2 - 2 q.v. */
int32_t tsh_probe_scan_keys(tsh_index *idx, const float *query, float *out_keys, float *out_eps_rel, float *out_delta_abs);
int32_t tsh_probe_batch_keys(tsh_index *idx, const float *queries, int32_t nq, int32_t k, float *out_keys,
                             float *out_delta2);
/* The band of the last tsh_probe_batch_keys call (same nq) row by row: the fp16 keys of an L2 / inner-product index are
 * known to alpha |v| + beta each -- the operand roundings act on the products, 2^-10 |q| |v| a row -- and the batched path
 * widens every row's key by its own band (a short row's far less than the longest row's).  out_alpha2[q] = 2 alpha_q,
 * out_beta2[q] = 2 * 1.0001 * beta_q: 2 |key - exact| <= out_alpha2[q] |v| + out_beta2[q] for every row v.  Key kernels
 * and metrics with one band for all rows report alpha = 0 and out_beta2 = out_delta2. */
int32_t tsh_probe_batch_row_band(tsh_index *idx, int32_t nq, float *out_alpha2, float *out_beta2);

/* Tuning knobs (no reference counterpart).  TSH_OPT_BATCH_MIN_NQ: when tsh_search /
 * tsh_search_shard answer a multi-query call on the batched matrix-core path:
 * 0 = never; 1 (default) = whenever its estimated cost is below that of nq pipelined
 * single-query scans (on 1 M x 768 rows from two queries on, on 100 k rows from about
 * six); n >= 2 = from n queries per call on.  Results are identical either way. */
#define TSH_OPT_BATCH_MIN_NQ 1
/* TSH_OPT_BATCH_KERNEL: how the batched path forms its pre-filter keys (the f64
 * rerank decides, and each variant's band covers its own error, so results are
 * identical whichever runs):
 *   0  f32 MFMA on the rows as stored;
 *   1  bf16x3: each f32 operand split into bf16 hi + lo, three bf16 MFMAs per
 *      product (error 3.1 * 2^-18 |q||v| on top of the f32 accumulation bound);
 *      costs a second copy of the rows, 4 B per element;
 *   2  f16: operands rounded to fp16 after an exact power-of-two scaling (cosine
 *      rows normalised first), one f16 MFMA per product (error 2^-10 |q||v|);
 *      costs a copy of 2 B per element;
 *   3  (default) auto: f16 for cosine indexes, whose keys are scale-free, for
 *      inner-product indexes (a row's band is its own: alpha |v| + beta) and for L2
 *      indexes while the band's shared term, ~ c (1 + max|v|^2), stays small against
 *      the spacing of the shortest rows' keys (~ 0.1 min|v|): norms a factor 32 apart
 *      are served by f16, norms 2^-6 .. 2^6 by bf16x3; an index whose f16 candidate
 *      lists overflow all the same is moved to bf16x3 for its next 256 batched calls.
 * The copy is built by the first batched search and kept current across appends. */
#define TSH_OPT_BATCH_KERNEL 2
/* TSH_OPT_EXCHANGE_AHEAD (process-wide; idx is ignored and may be NULL; default 0): 1 = tsh_search_sharded over RCCL
 * launches a group's all-gather as soon as the group's scans are ENQUEUED, ordered on the communicator's stream behind
 * the events of the group's last block writers, instead of when the host has seen the blocks final -- the collective's
 * launch (tens of microseconds of host time) then hides behind the scans.  Blocks carry a generation
 * (their header's pad[1]) so that a block of an earlier call is never taken for an answer, and a block the host is
 * still going to redo (ties) asks for a retry like a truncated one.  Measured on one MI355X with 125 k x 768 shards it
 * LOSES 3-6 % (the early packets on the communicator's queue run late), hence off by default;
 * kept for hosts whose collectives are costlier to launch.  Same value on every rank. */
#define TSH_OPT_EXCHANGE_AHEAD 3
/* TSH_OPT_EXACT_SCAN_ROWS (default and maximum 16384; 0 = never): a single-query search that has at most this many
 * rows to look at -- the kept rows of a selective row mask, or all rows of a small index or shard -- computes the exact
 * f64 sums of ALL of them in one launch (the re-rank's arithmetic, one wave per few rows) and selects the k smallest
 * exact distances in a second: the f32 pre-filter has nothing to spare there, and its three dependent dispatches
 * (scan, select, re-rank) are what such a search waits for.  Results are identical either way (the candidates are then
 * exactly the k winners); counted in tsh_counters.exact_scans. */
#define TSH_OPT_EXACT_SCAN_ROWS 4
/* TSH_OPT_EXACT_SELECT (default 1): what follows the exact scan of such a search.  1 = the wide pick: the scan leaves
 * every wave's smallest distance, the k-th smallest of those bounds the k-th distance from above, and one workgroup per
 * 256 rows emits every row at or below the bound -- the k winners plus a few rows that cannot win, no ranking on the
 * device (the finaliser orders and cuts as always); a bound that lets in more rows than the block holds (ties by the
 * hundred) is finished by the one-workgroup select, and searches with fewer than 16 k rows to look at take that select
 * directly.  0 = always the select (round 5: exactly min(k, live rows) entries, one compute unit, 9-17 us for 2 k-10 k
 * rows).  Results are identical. */
#define TSH_OPT_EXACT_SELECT 5
/* TSH_OPT_BATCH_HUB (default 0): 1 = batched searches of an L2 / inner-product index on fp16 keys also score the index's
 * HUB rows densely -- the 4096 rows whose norm alone puts them near every query: the shortest (L2) resp. the longest
 * (inner product), kept as a gathered fp16 copy -- and take the smaller of two thresholds: the sample's (an estimate,
 * verified) and the hub rows' k-th smallest key (a bound by construction).  Measured on one MI355X it thins the key
 * kernel's epilogue on corpora whose neighbours are their short rows (key passes - 3 to - 5 %) and costs the call as
 * much as it saves: off unless a host knows its corpus to be that kind.  Results are identical. */
#define TSH_OPT_BATCH_HUB 6
/* TSH_OPT_BATCH_GROUP (default 1): 1 = the fp16 copy of an L2 / inner-product index that batched searches score holds
 * its rows ordered by norm inside blocks of 8192 consecutive rows (the rows every query passes -- the short ones under
 * L2, the long ones under inner product -- then share a few tiles of the key kernel instead of slowing a quarter of
 * them down); 0 = in row order.  Results are identical either way; the copy is rebuilt by the next batched search
 * after a change. */
#define TSH_OPT_BATCH_GROUP 7
/* TSH_OPT_TEST_HOOKS (process-wide; idx is ignored and may be NULL): value TSH_TEST_HOOKS_MAGIC switches the
 * library's TEST hooks on, 0 off.  Only then does it read the environment variables that change what it loads or make
 * it fail on purpose -- TSH_RCCL_LIB (a stand-in for librccl: tests/fake_rccl), TSH_TEST_FAIL_ALLOC_OVER (device
 * allocations fail), TSH_SHARDS_SHARE_DEVICES (several shards of one handle on one GPU).  Without the call those
 * variables change nothing: an embedded database must not be steerable through its environment.  Experiment
 * switches (kernel shapes, stream layouts) exist in probe builds only (-DTSH_PROBES).  A release build reads
 * LOCAL_WORLD_SIZE, TSH_BLOCKING_WAIT, TSH_HOST_THREADS (how the host side waits: DESIGN.md appendix B) and
 * TSH_TRACE_BATCH (stderr diagnostics), nothing else. */
#define TSH_OPT_TEST_HOOKS 1000
#define TSH_TEST_HOOKS_MAGIC 0x7465737468ll /* "testh" */
int32_t tsh_index_set_option(tsh_index *idx, int32_t option, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* TOSTORE_HIP_H */
