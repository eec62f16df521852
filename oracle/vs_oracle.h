/*
 * vs_oracle.h -- CPU restatement of ToStore's exact vector-search arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference (tocreator/tostore v3.2.0, pure Dart) ships
 * no test, golden vector or fixture for vectorSearch, and no Dart SDK exists
 * in the build image, so this restatement could not be checked against the
 * reference's own outputs.  It is pinned instead by (i) hand-derivable cases,
 * (ii) the README/example vectors, and (iii) bit-for-bit agreement with a
 * second, independently written NumPy restatement (oracle/np_oracle.py).
 *
 * All "ref:" citations are paths under /root/reference/lib/src/.
 */
#ifndef VS_ORACLE_H
#define VS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* metric codes = enum order of VectorDistanceMetric, ref: model/table_schema.dart:2511-2531 */
#define VSO_L2 0
#define VSO_IP 1
#define VSO_COSINE 2

/* A1  ref: core/vector_index_manager.dart:1385-1392 (query),
 *          core/compute/vector_batch_prepare_compute.dart:79-86 (stored rows) */
void vso_to_float32(const double *values, int64_t len, int dim, float *out);

/* A2  ref: core/vector_index_manager.dart:1395-1408.  out may alias v. */
void vso_normalize_f32(const float *v, int dim, float *out);

/* A3-A5  ref: core/ngh_graph_engine.dart:920-946 */
double vso_l2_distance(const float *a, const float *b, int d);
double vso_inner_product(const float *a, const float *b, int d);
double vso_cosine_similarity(const float *a, const float *b, int d);
/* ref: core/ngh_graph_engine.dart:908-918 */
double vso_exact_distance(const float *a, const float *b, int d, int metric);

/* the raw f64 accumulations of :920-946 before sqrt / divide (tests build
 * candidate blocks from them): L2 s0 = sum (a-b)^2; IP s0 = sum a*b;
 * cosine s0 = dot, s1 = sum b*b */
void vso_exact_sums(const float *a, const float *b, int d, int metric, double *s0, double *s1);

/* A8  ref: core/vector_index_manager.dart:1411-1423 */
double vso_distance_to_score(double distance, int metric);

/* Dart double.compareTo ([external] Dart SDK semantics): -1/0/+1; NaN is the
 * greatest value and equal to itself; -0.0 < +0.0. */
int vso_compare_double(double a, double b);

/* A6 applied exhaustively ("the reference with ef -> infinity"):
 * for every row i in [0,n) with keep bit set (keep == NULL: all rows):
 *   dist = exact_distance(query,row_i); drop if dist > threshold (strict;
 *   threshold NaN = none)            ref: core/ngh_graph_engine.dart:122-130
 * sort ascending by compareTo, ties by row id ascending (the reference leaves
 * tie order open: List.sort is unstable), keep first k   ref: :133-134
 * keep: bit i of byte i/8 (LSB first), 1 = row is live.
 * Returns the number of results written (<= k). */
int64_t vso_search_exhaustive(const float *rows, int64_t n, int d, int metric,
                              const float *query, int64_t k, double threshold,
                              const uint8_t *keep, int64_t *out_ids,
                              double *out_dist);

/* Same result, single pass with a bounded heap (used as the timed CPU
 * baseline; vso_search_exhaustive sorts all n distances). */
int64_t vso_search_heap(const float *rows, int64_t n, int d, int metric,
                        const float *query, int64_t k, double threshold,
                        const uint8_t *keep, int64_t *out_ids,
                        double *out_dist);

/* All n distances (no threshold / sort), for property tests. */
void vso_all_distances(const float *rows, int64_t n, int d, int metric,
                       const float *query, double *out_dist);

/* N4 (write path): batch PQ encode.  ref: core/compute_tasks.dart:2292-2326
 * (batchPqEncode) == core/vector_quantizer.dart:357-368,461-483 (encode /
 * _nearestCentroid): per vector and sub-space m, the index of the first
 * centroid with the smallest f64-accumulated squared distance (strict `<`, so
 * ties and NaN keep the lower index).  codebook layout :15
 * centroids[(m*K + k)*subDim + d]; vectors n x dim with dim >= subspaces*subDim;
 * codes n x subspaces. */
void vso_pq_encode(const float *codebook, int subspaces, int centroids, int sub_dim,
                   const float *vectors, int64_t n, int dim, uint8_t *codes);

/* N4: PQ codebook training for ONE sub-space, given the k sample indices the
 * reference draws with Random(42 + subspaceIndex).nextInt(n) (Dart PRNG, not
 * restated: the caller supplies them).  ref: core/compute_tasks.dart:2135-2266
 * (trainPqSubspace).  Mixed arithmetic restated exactly: centroid norms f64
 * accumulate -> *0.5 -> f32; assignment maximises dot - norm with, when
 * subDim % 4 == 0, Float32x4 products (f32 multiply) summed ((x+y)+z)+w in f64,
 * else f64 products; update accumulates sums in a Float32List (f32 rounding per
 * add, sample order), divides by count in f64, stores f32; stops after the
 * update of the first iteration in which no coordinate moved by > 1e-4.
 * sub_samples n x sub_dim; init_index k; out_centroids k x sub_dim. */
void vso_pq_train_subspace(const float *sub_samples, int64_t n, int sub_dim, int k, int iterations,
                           const int32_t *init_index, float *out_centroids);
/* N3: the k-means++ trainer the reference uses for first batches of fewer than 100 vectors
 * (ref: core/vector_quantizer.dart:81-350); `first` and u[k - 1] stand for Dart's Random(42) draws */
void vso_pq_train_subspace_pp(const float *sub_samples, int64_t n, int sub_dim, int k, int iterations, int32_t first,
                              const double *u, float *centroids);

/* A7  page framing.  ref: core/btree_page.dart:61-89 (CRC32 IEEE, reflected,
 * poly 0xEDB88320), :132-234 (20-byte 'TPG2' header) */
uint32_t vso_crc32(const uint8_t *data, size_t len);

#define VSO_PAGE_HEADER_SIZE 20
#define VSO_PAGE_TYPE_NGH_META 5       /* ref: core/btree_page.dart:14-55 enum index */
#define VSO_PAGE_TYPE_NGH_RAWVECTOR 8

/* ref: core/ngh_page.dart:575-579 */
int vso_vectors_per_raw_page(int page_size, int dimensions, int bpe);

/* Build one raw-vector page (ref: core/ngh_page.dart:418-429 payload,
 * core/btree_page.dart:188-213 framing).  vectors: count x dims float32.
 * precision: 0=f64 1=f32 2=i8 (ref: core/ngh_page.dart:397-416 conversions).
 * out must hold page_size bytes.  Returns 0 ok, -1 overflow. */
int vso_rawvec_page_build(const float *vectors, int count, int dims,
                          int precision, int page_size, uint8_t *out);

/* Parse one raw-vector page into float32 (ref: core/ngh_page.dart:431-450,
 * :364-391; core/btree_page.dart:215-233).  Returns vector count, or -1 for
 * an invalid page: the reference THROWS on a bad magic / CRC / length
 * (btree_page.dart:215-233) and substitutes an all-zero page when the file
 * is short or the payload does not decode (ngh_partition_manager.dart:276-295).
 * out_vectors must hold max_vectors x dims floats. */
int vso_rawvec_page_parse(const uint8_t *page, int page_size, int dims,
                          float *out_vectors, int max_vectors,
                          int *out_precision);

/* NGH partition meta page (pageNo 0).  ref: core/ngh_page.dart:29-98 */
int vso_ngh_meta_page_build(int partition_no, int data_category,
                            int64_t total_entries, int64_t file_size,
                            int page_size, uint8_t *out);

/* nodeId -> (partition, pageNo, slot).  ref: model/ngh_index_meta.dart:480-490;
 * pages_per_partition = maxPartitionFileSize ~/ nghPageSize (:163,178), data
 * pages start at pageNo 1 (:232) */
void vso_rawvec_locate(int64_t node_id, int vectors_per_page,
                       int64_t pages_per_partition, int64_t *partition,
                       int64_t *page_no, int *slot);

#ifdef __cplusplus
}
#endif
#endif
