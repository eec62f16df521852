/*
 * vs_oracle.c -- CPU restatement of ToStore's exact vector-search arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY (see vs_oracle.h).  PARITY UNPINNED by the
 * reference's own tests (it has none for this path); see header.
 *
 * Build with -ffp-contract=off: Dart never fuses a*b+c, so every multiply and
 * add below must round separately in IEEE binary64.
 *
 * "ref:" = path under /root/reference/lib/src/.
 */
#include "vs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- A1: List<double> -> Float32List(dimensions) ------------------------
 * ref: core/vector_index_manager.dart:1385-1392,
 *      core/compute/vector_batch_prepare_compute.dart:79-86
 * truncate or zero-pad to `dim`; Float32List store rounds f64->f32 RNE. */
void vso_to_float32(const double *values, int64_t len, int dim, float *out) {
  int64_t copy = len < dim ? len : dim;
  int64_t i;
  for (i = 0; i < dim; i++) out[i] = 0.0f;
  for (i = 0; i < copy; i++) out[i] = (float)values[i];
}

/* ---- A2: query normalisation (cosine only) -------------------------------
 * ref: core/vector_index_manager.dart:1395-1408
 * mag accumulates in f64 over widened f32 elements; zero magnitude returns
 * the input unchanged; result[i] = f32(v[i] * (1.0/mag)). */
void vso_normalize_f32(const float *v, int dim, float *out) {
  double mag = 0;
  int i;
  for (i = 0; i < dim; i++) mag += (double)v[i] * (double)v[i];
  mag = sqrt(mag);
  if (mag == 0) {
    if (out != v) memcpy(out, v, (size_t)dim * sizeof(float));
    return;
  }
  {
    double inv = 1.0 / mag;
    for (i = 0; i < dim; i++) out[i] = (float)((double)v[i] * inv);
  }
}

/* ---- A3: ref: core/ngh_graph_engine.dart:920-927 ------------------------ */
double vso_l2_distance(const float *a, const float *b, int d) {
  double sum = 0;
  int i;
  for (i = 0; i < d; i++) {
    double diff = (double)a[i] - (double)b[i];
    sum += diff * diff;
  }
  return sqrt(sum);
}

/* ---- A4: ref: core/ngh_graph_engine.dart:929-935 ------------------------ */
double vso_inner_product(const float *a, const float *b, int d) {
  double sum = 0;
  int i;
  for (i = 0; i < d; i++) sum += (double)a[i] * (double)b[i];
  return sum;
}

/* ---- A5: ref: core/ngh_graph_engine.dart:937-946 ------------------------ */
double vso_cosine_similarity(const float *a, const float *b, int d) {
  double dot = 0, magA = 0, magB = 0, denom;
  int i;
  for (i = 0; i < d; i++) {
    dot += (double)a[i] * (double)b[i];
    magA += (double)a[i] * (double)a[i];
    magB += (double)b[i] * (double)b[i];
  }
  denom = sqrt(magA) * sqrt(magB);
  return denom > 0 ? dot / denom : 0;
}

/* ---- ref: core/ngh_graph_engine.dart:908-918 ---------------------------- */
double vso_exact_distance(const float *a, const float *b, int d, int metric) {
  switch (metric) {
    case VSO_L2:
      return vso_l2_distance(a, b, d);
    case VSO_IP:
      return -vso_inner_product(a, b, d);
    default:
      return 1.0 - vso_cosine_similarity(a, b, d);
  }
}

void vso_exact_sums(const float *a, const float *b, int d, int metric, double *s0, double *s1) {
  double x = 0, y = 0;
  int i;
  for (i = 0; i < d; i++) {
    if (metric == VSO_L2) {
      double diff = (double)a[i] - (double)b[i];
      x += diff * diff;
    } else {
      x += (double)a[i] * (double)b[i];
      if (metric == VSO_COSINE) y += (double)b[i] * (double)b[i];
    }
  }
  *s0 = x;
  *s1 = y;
}

/* ---- Dart double.compareTo [external: Dart SDK] ------------------------- */
int vso_compare_double(double a, double b) {
  if (a < b) return -1;
  if (a > b) return 1;
  if (a == b) {
    if (a == 0.0) {
      int an = signbit(a) != 0, bn = signbit(b) != 0;
      if (an == bn) return 0;
      return an ? -1 : 1;
    }
    return 0;
  }
  if (isnan(a)) return isnan(b) ? 0 : 1;
  return -1;
}

/* ---- A8: ref: core/vector_index_manager.dart:1411-1423 -------------------
 * cosine uses num.clamp(0.0,1.0), which compares with compareTo: NaN -> 1.0,
 * -0.0 -> 0.0 [external: Dart SDK num.clamp]. */
double vso_distance_to_score(double distance, int metric) {
  switch (metric) {
    case VSO_L2:
      return 1.0 / (1.0 + distance);
    case VSO_IP:
      return 1.0 / (1.0 + exp(-(-distance)));
    default: {
      double s = 1.0 - distance;
      if (vso_compare_double(s, 0.0) < 0) return 0.0;
      if (vso_compare_double(s, 1.0) > 0) return 1.0;
      return s;
    }
  }
}

/* ---- A6 exhaustive ------------------------------------------------------ */
typedef struct {
  double dist;
  int64_t id;
} vso_hit;

static int hit_cmp(const void *pa, const void *pb) {
  const vso_hit *a = (const vso_hit *)pa, *b = (const vso_hit *)pb;
  int c = vso_compare_double(a->dist, b->dist);
  if (c) return c;
  return a->id < b->id ? -1 : (a->id > b->id ? 1 : 0);
}

static int keep_bit(const uint8_t *keep, int64_t i) {
  return keep == NULL || ((keep[i >> 3] >> (i & 7)) & 1);
}

void vso_all_distances(const float *rows, int64_t n, int d, int metric,
                       const float *query, double *out_dist) {
  int64_t i;
  for (i = 0; i < n; i++)
    out_dist[i] = vso_exact_distance(query, rows + i * (int64_t)d, d, metric);
}

int64_t vso_search_exhaustive(const float *rows, int64_t n, int d, int metric,
                              const float *query, int64_t k, double threshold,
                              const uint8_t *keep, int64_t *out_ids,
                              double *out_dist) {
  vso_hit *hits;
  int64_t i, m = 0, r;
  if (n <= 0 || k <= 0) return 0;
  hits = (vso_hit *)malloc((size_t)n * sizeof(vso_hit));
  if (!hits) return -1;
  for (i = 0; i < n; i++) {
    double dist;
    if (!keep_bit(keep, i)) continue;
    /* ref: ngh_graph_engine.dart:126 -- query is `a`, stored row is `b` */
    dist = vso_exact_distance(query, rows + i * (int64_t)d, d, metric);
    /* ref: :127 `distanceThreshold != null && exactDist > distanceThreshold` */
    if (!isnan(threshold) && dist > threshold) continue;
    hits[m].dist = dist;
    hits[m].id = i;
    m++;
  }
  qsort(hits, (size_t)m, sizeof(vso_hit), hit_cmp); /* ref: :133 */
  r = m < k ? m : k;                                /* ref: :134 */
  for (i = 0; i < r; i++) {
    out_ids[i] = hits[i].id;
    out_dist[i] = hits[i].dist;
  }
  free(hits);
  return r;
}

/* bounded max-heap on (dist,id) under hit_cmp; root = worst kept hit */
static void sift_down(vso_hit *h, int64_t n, int64_t i) {
  for (;;) {
    int64_t l = 2 * i + 1, r = l + 1, m = i;
    vso_hit t;
    if (l < n && hit_cmp(&h[l], &h[m]) > 0) m = l;
    if (r < n && hit_cmp(&h[r], &h[m]) > 0) m = r;
    if (m == i) return;
    t = h[i];
    h[i] = h[m];
    h[m] = t;
    i = m;
  }
}

int64_t vso_search_heap(const float *rows, int64_t n, int d, int metric,
                        const float *query, int64_t k, double threshold,
                        const uint8_t *keep, int64_t *out_ids,
                        double *out_dist) {
  vso_hit *heap;
  int64_t i, m = 0;
  if (n <= 0 || k <= 0) return 0;
  heap = (vso_hit *)malloc((size_t)k * sizeof(vso_hit));
  if (!heap) return -1;
  for (i = 0; i < n; i++) {
    vso_hit h;
    if (!keep_bit(keep, i)) continue;
    h.dist = vso_exact_distance(query, rows + i * (int64_t)d, d, metric);
    h.id = i;
    if (!isnan(threshold) && h.dist > threshold) continue;
    if (m < k) {
      int64_t c = m++;
      heap[c] = h;
      while (c > 0) { /* sift up */
        int64_t p = (c - 1) / 2;
        vso_hit t;
        if (hit_cmp(&heap[c], &heap[p]) <= 0) break;
        t = heap[c];
        heap[c] = heap[p];
        heap[p] = t;
        c = p;
      }
    } else if (hit_cmp(&h, &heap[0]) < 0) {
      heap[0] = h;
      sift_down(heap, m, 0);
    }
  }
  qsort(heap, (size_t)m, sizeof(vso_hit), hit_cmp);
  for (i = 0; i < m; i++) {
    out_ids[i] = heap[i].id;
    out_dist[i] = heap[i].dist;
  }
  free(heap);
  return m;
}

/* ---- N4: ref: core/compute_tasks.dart:2292-2326 ------------------------------ */
void vso_pq_encode(const float *codebook, int subspaces, int centroids, int sub_dim,
                   const float *vectors, int64_t n, int dim, uint8_t *codes) {
  int64_t v;
  for (v = 0; v < n; v++) {
    const float *vec = vectors + v * (int64_t)dim;
    int m;
    for (m = 0; m < subspaces; m++) {
      int sub_start = m * sub_dim, best_idx = 0, c, d;
      double best = INFINITY;
      const float *cb = codebook + (int64_t)m * centroids * sub_dim;
      for (c = 0; c < centroids; c++) {
        double dist = 0;
        const float *cc = cb + (int64_t)c * sub_dim;
        for (d = 0; d < sub_dim; d++) {
          double diff = (double)vec[sub_start + d] - (double)cc[d];
          dist += diff * diff;
        }
        if (dist < best) {
          best = dist;
          best_idx = c;
        }
      }
      codes[v * subspaces + m] = (uint8_t)best_idx; /* code[m] = bestIdx into a Uint8List */
    }
  }
}

/* ---- N4: ref: core/compute_tasks.dart:2135-2266 (trainPqSubspace) ------------ */
void vso_pq_train_subspace(const float *data, int64_t n, int sub_dim, int k, int iterations,
                           const int32_t *init_index, float *centroids) {
  int32_t *assign = (int32_t *)malloc((size_t)n * sizeof(int32_t));
  int32_t *counts = (int32_t *)malloc((size_t)k * sizeof(int32_t));
  float *sums = (float *)malloc((size_t)k * sub_dim * sizeof(float));
  float *norms = (float *)malloc((size_t)k * sizeof(float));
  int use_simd = (sub_dim % 4 == 0), iter, c, d;
  int64_t i;
  for (c = 0; c < k; c++) /* :2144-2151 */
    for (d = 0; d < sub_dim; d++) centroids[c * sub_dim + d] = data[(int64_t)init_index[c] * sub_dim + d];
  for (iter = 0; iter < iterations; iter++) {
    int changed = 0;
    for (c = 0; c < k; c++) { /* :2166-2174 */
      double norm = 0;
      for (d = 0; d < sub_dim; d++) {
        double val = (double)centroids[c * sub_dim + d];
        norm += val * val;
      }
      norms[c] = (float)(norm * 0.5);
    }
    for (i = 0; i < n; i++) { /* :2183-2233 */
      double best = -INFINITY;
      int best_idx = 0;
      for (c = 0; c < k; c++) {
        double dot = 0, score;
        if (use_simd) {
          for (d = 0; d < sub_dim; d += 4) { /* Float32x4 multiply: four f32 products */
            float rx = data[i * sub_dim + d] * centroids[c * sub_dim + d];
            float ry = data[i * sub_dim + d + 1] * centroids[c * sub_dim + d + 1];
            float rz = data[i * sub_dim + d + 2] * centroids[c * sub_dim + d + 2];
            float rw = data[i * sub_dim + d + 3] * centroids[c * sub_dim + d + 3];
            dot += (((double)rx + (double)ry) + (double)rz) + (double)rw;
          }
        } else {
          for (d = 0; d < sub_dim; d++) dot += (double)data[i * sub_dim + d] * (double)centroids[c * sub_dim + d];
        }
        score = dot - (double)norms[c];
        if (score > best) {
          best = score;
          best_idx = c;
        }
      }
      assign[i] = best_idx;
    }
    memset(sums, 0, (size_t)k * sub_dim * sizeof(float)); /* :2236-2248 */
    memset(counts, 0, (size_t)k * sizeof(int32_t));
    for (i = 0; i < n; i++) {
      c = assign[i];
      counts[c]++;
      for (d = 0; d < sub_dim; d++)
        sums[c * sub_dim + d] = (float)((double)sums[c * sub_dim + d] + (double)data[i * sub_dim + d]);
    }
    for (c = 0; c < k; c++) { /* :2250-2261 */
      double inv;
      if (counts[c] == 0) continue;
      inv = 1.0 / (double)counts[c];
      for (d = 0; d < sub_dim; d++) {
        double new_val = (double)sums[c * sub_dim + d] * inv;
        if (fabs((double)centroids[c * sub_dim + d] - new_val) > 1e-4) changed = 1;
        centroids[c * sub_dim + d] = (float)new_val;
      }
    }
    if (!changed) break; /* :2262 */
  }
  free(assign);
  free(counts);
  free(sums);
  free(norms);
}

/* ---- N3: the reference's OTHER trainer, for first batches of fewer than 100 vectors.
 * ref: core/vector_quantizer.dart:81-350 (VectorQuantizer.train), chosen at core/vector_index_manager.dart:744,842-849.
 * One sub-space: k-means++ seeding, then Lloyd iterations with squared distances (the isolate trainer above ranks by
 * dot - |c|^2 / 2 and starts from k random samples instead).  Dart's Random(42) -- one stream shared by all
 * sub-spaces, nextInt(n) then k - 1 nextDouble() per sub-space -- is not reproducible without a Dart runtime, so the
 * caller supplies the draws: first = the nextInt result, u[c - 1] = the nextDouble that picks centroid c.
 * Arithmetic widths as in the Dart text: Float32x4 lanes subtract and multiply in f32 and are summed as doubles
 * (:143-146); the scalar path (sub_dim % 4 != 0) subtracts the widened values in f64 (:262-266). */
static double vso_pp_dist(const float *a, const float *b, int sub_dim, int use_simd) {
  double dist = 0;
  int d;
  if (use_simd) {
    for (d = 0; d < sub_dim; d += 4) {
      float dx = a[d] - b[d], dy = a[d + 1] - b[d + 1], dz = a[d + 2] - b[d + 2], dw = a[d + 3] - b[d + 3];
      float mx = dx * dx, my = dy * dy, mz = dz * dz, mw = dw * dw;
      dist += (((double)mx + (double)my) + (double)mz) + (double)mw;
    }
  } else {
    for (d = 0; d < sub_dim; d++) {
      double diff = (double)a[d] - (double)b[d];
      dist += diff * diff;
    }
  }
  return dist;
}

void vso_pq_train_subspace_pp(const float *data, int64_t n, int sub_dim, int k, int iterations, int32_t first,
                              const double *u, float *centroids) {
  int32_t *assign = (int32_t *)malloc((size_t)n * sizeof(int32_t));
  int32_t *counts = (int32_t *)malloc((size_t)k * sizeof(int32_t));
  float *sums = (float *)malloc((size_t)k * sub_dim * sizeof(float));
  double *min_d = (double *)malloc((size_t)n * sizeof(double));
  int use_simd = (sub_dim % 4 == 0), iter, c, d;
  int64_t i;
  for (d = 0; d < sub_dim; d++) centroids[d] = data[(int64_t)first * sub_dim + d]; /* :126-130 */
  for (i = 0; i < n; i++) min_d[i] = INFINITY;
  for (c = 1; c < k; c++) { /* :135-172 */
    double total = 0;
    int64_t selected = n - 1;
    for (i = 0; i < n; i++) {
      double dist = vso_pp_dist(data + i * sub_dim, centroids + (int64_t)(c - 1) * sub_dim, sub_dim, use_simd);
      if (dist < min_d[i]) min_d[i] = dist;
      total += min_d[i];
    }
    if (total > 0) {
      double threshold = u[c - 1] * total;
      for (i = 0; i < n; i++) {
        threshold -= min_d[i];
        if (threshold <= 0) {
          selected = i;
          break;
        }
      }
    }
    for (d = 0; d < sub_dim; d++) centroids[(int64_t)c * sub_dim + d] = data[selected * sub_dim + d];
  }
  for (iter = 0; iter < iterations; iter++) { /* :180-232 */
    int changed = 0;
    for (i = 0; i < n; i++) {
      int best_idx = 0;
      double best = INFINITY;
      for (c = 0; c < k; c++) {
        double dist = vso_pp_dist(data + i * sub_dim, centroids + (int64_t)c * sub_dim, sub_dim, use_simd);
        if (dist < best) {
          best = dist;
          best_idx = c;
        }
      }
      assign[i] = best_idx;
    }
    memset(sums, 0, (size_t)k * sub_dim * sizeof(float));
    memset(counts, 0, (size_t)k * sizeof(int32_t));
    for (i = 0; i < n; i++) {
      c = assign[i];
      counts[c]++;
      for (d = 0; d < sub_dim; d++)
        sums[c * sub_dim + d] = (float)((double)sums[c * sub_dim + d] + (double)data[i * sub_dim + d]);
    }
    for (c = 0; c < k; c++) {
      double inv;
      if (counts[c] == 0) continue;
      inv = 1.0 / (double)counts[c];
      for (d = 0; d < sub_dim; d++) {
        double new_val = (double)sums[c * sub_dim + d] * inv;
        if (fabs((double)centroids[c * sub_dim + d] - new_val) > 1e-6) changed = 1;
        centroids[c * sub_dim + d] = (float)new_val;
      }
    }
    if (!changed) break;
  }
  free(assign);
  free(counts);
  free(sums);
  free(min_d);
}

/* ---- A7: CRC32.  ref: core/btree_page.dart:61-89 ------------------------ */
uint32_t vso_crc32(const uint8_t *data, size_t len) {
  static uint32_t table[256];
  static int ready = 0;
  uint32_t c;
  size_t i;
  if (!ready) {
    uint32_t n;
    for (n = 0; n < 256; n++) {
      int k;
      c = n;
      for (k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
      table[n] = c;
    }
    ready = 1;
  }
  c = 0xFFFFFFFFu;
  for (i = 0; i < len; i++) c = table[(c ^ data[i]) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

static void put_u16(uint8_t *p, uint32_t v) {
  p[0] = (uint8_t)v;
  p[1] = (uint8_t)(v >> 8);
}
static void put_u32(uint8_t *p, uint32_t v) {
  p[0] = (uint8_t)v;
  p[1] = (uint8_t)(v >> 8);
  p[2] = (uint8_t)(v >> 16);
  p[3] = (uint8_t)(v >> 24);
}
static void put_u64(uint8_t *p, uint64_t v) {
  put_u32(p, (uint32_t)v);
  put_u32(p + 4, (uint32_t)(v >> 32));
}
static uint32_t get_u16(const uint8_t *p) { return p[0] | ((uint32_t)p[1] << 8); }
static uint32_t get_u32(const uint8_t *p) {
  return p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) |
         ((uint32_t)p[3] << 24);
}

/* ref: core/ngh_page.dart:575-579 (usable = pageSize - 20 - 8 - 64) */
int vso_vectors_per_raw_page(int page_size, int dimensions, int bpe) {
  int usable = page_size - VSO_PAGE_HEADER_SIZE - 8 - 64;
  int vec = dimensions * bpe;
  return (usable > 0 && vec > 0) ? usable / vec : 0;
}

static int bpe_of(int precision) { return precision == 0 ? 8 : (precision == 2 ? 1 : 4); }

/* ref: core/btree_page.dart:148-158 header encode, :188-213 buildPageBytes */
static int frame_page(int page_type, const uint8_t *payload, uint32_t payload_len,
                      int page_size, uint8_t *out) {
  if (VSO_PAGE_HEADER_SIZE + (int64_t)payload_len > page_size) return -1;
  memset(out, 0, (size_t)page_size);
  put_u32(out + 0, 0x32475054u); /* 'TPG2' little-endian */
  put_u16(out + 4, VSO_PAGE_HEADER_SIZE);
  out[6] = (uint8_t)page_type;
  out[7] = 0; /* flags */
  put_u32(out + 8, payload_len);
  put_u32(out + 12, vso_crc32(payload, payload_len));
  put_u32(out + 16, 0);
  memcpy(out + VSO_PAGE_HEADER_SIZE, payload, payload_len);
  return 0;
}

/* Dart double.round(): half away from zero */
static double dart_round(double x) { return x < 0 ? -floor(-x + 0.5) : floor(x + 0.5); }

int vso_rawvec_page_build(const float *vectors, int count, int dims,
                          int precision, int page_size, uint8_t *out) {
  int bpe = bpe_of(precision);
  size_t data_len = (size_t)count * (size_t)dims * (size_t)bpe;
  size_t total = 8 + data_len;
  uint8_t *payload = (uint8_t *)calloc(1, total ? total : 1);
  int rc;
  size_t i;
  if (!payload) return -1;
  /* ref: core/ngh_page.dart:418-429 */
  put_u16(payload + 0, (uint32_t)count);
  put_u16(payload + 2, (uint32_t)dims);
  payload[4] = (uint8_t)precision;
  for (i = 0; i < (size_t)count * (size_t)dims; i++) {
    uint8_t *p = payload + 8 + i * (size_t)bpe;
    float f = vectors[i];
    if (precision == 1) { /* ref: :399-401 */
      uint32_t u;
      memcpy(&u, &f, 4);
      put_u32(p, u);
    } else if (precision == 0) { /* ref: :403-405 */
      double dv = (double)f;
      uint64_t u;
      memcpy(&u, &dv, 8);
      put_u64(p, u);
    } else { /* ref: :408-412  clamp [-1,1] (NaN -> 1.0 via compareTo), *127, round */
      double c = (double)f;
      if (vso_compare_double(c, -1.0) < 0) c = -1.0;
      else if (vso_compare_double(c, 1.0) > 0) c = 1.0;
      *p = (uint8_t)(int8_t)(int)dart_round(c * 127.0);
    }
  }
  rc = frame_page(VSO_PAGE_TYPE_NGH_RAWVECTOR, payload, (uint32_t)total, page_size, out);
  free(payload);
  return rc;
}

int vso_rawvec_page_parse(const uint8_t *page, int page_size, int dims,
                          float *out_vectors, int max_vectors,
                          int *out_precision) {
  uint32_t plen, crc, vcount, pdims;
  const uint8_t *payload;
  int prec, bpe;
  size_t i, nelem;
  /* ref: core/btree_page.dart:162-183 tryDecode */
  if (page_size < VSO_PAGE_HEADER_SIZE) return -1;
  if (get_u32(page) != 0x32475054u) return -1;
  if (get_u16(page + 4) != VSO_PAGE_HEADER_SIZE) return -1;
  if (page[6] >= 10) return -1; /* BTreePageType.values.length */
  plen = get_u32(page + 8);
  crc = get_u32(page + 12);
  /* ref: :215-233 parsePageBytes */
  if ((int64_t)VSO_PAGE_HEADER_SIZE + plen > (int64_t)page_size) return -1;
  payload = page + VSO_PAGE_HEADER_SIZE;
  if (vso_crc32(payload, plen) != crc) return -1;
  /* ref: core/ngh_page.dart:431-450 tryDecodePayload (page type itself is
   * not checked by the reader: ngh_partition_manager.dart:284-286) */
  if (plen < 8) return -1;
  vcount = get_u16(payload);
  pdims = get_u16(payload + 2);
  prec = payload[4];
  if (pdims == 0) return -1;
  bpe = bpe_of(prec);
  if ((uint64_t)plen < 8 + (uint64_t)vcount * pdims * (uint64_t)bpe) return -1;
  if ((int)pdims != dims) return -1; /* caller contract: dims from meta.json */
  if ((int)vcount > max_vectors) return -1;
  if (out_precision) *out_precision = prec;
  nelem = (size_t)vcount * pdims;
  for (i = 0; i < nelem; i++) { /* ref: :364-391 getVectorAsFloat32 */
    const uint8_t *p = payload + 8 + i * (size_t)bpe;
    if (prec == 1) {
      uint32_t u = get_u32(p);
      memcpy(&out_vectors[i], &u, 4);
    } else if (prec == 0) {
      uint64_t u = (uint64_t)get_u32(p) | ((uint64_t)get_u32(p + 4) << 32);
      double dv;
      memcpy(&dv, &u, 8);
      out_vectors[i] = (float)dv;
    } else {
      out_vectors[i] = (float)((double)(int8_t)*p / 127.0);
    }
  }
  return (int)vcount;
}

/* ref: core/ngh_page.dart:62-75 */
int vso_ngh_meta_page_build(int partition_no, int data_category,
                            int64_t total_entries, int64_t file_size,
                            int page_size, uint8_t *out) {
  uint8_t payload[128];
  memset(payload, 0, sizeof payload);
  put_u32(payload + 0, 0x3148474Eu); /* 'NGH1' */
  put_u16(payload + 4, 1);
  put_u16(payload + 6, (uint32_t)data_category);
  put_u32(payload + 8, (uint32_t)partition_no);
  put_u32(payload + 12, 0);
  put_u64(payload + 16, (uint64_t)total_entries);
  put_u64(payload + 24, (uint64_t)file_size);
  put_u32(payload + 32, (uint32_t)-1); /* freeListHeadPageNo */
  put_u32(payload + 36, 0);
  return frame_page(VSO_PAGE_TYPE_NGH_META, payload, 128, page_size, out);
}

/* ref: model/ngh_index_meta.dart:480-490 with
 * rawVectorPagesPerPartition = maxPartitionFileSize ~/ nghPageSize (:163,178)
 * and firstDataPageNo = 1 (:232) */
void vso_rawvec_locate(int64_t node_id, int vectors_per_page,
                       int64_t pages_per_partition, int64_t *partition,
                       int64_t *page_no, int *slot) {
  int64_t logical = node_id / vectors_per_page;
  *partition = logical / pages_per_partition;
  *page_no = 1 + (logical % pages_per_partition);
  *slot = (int)(node_id % vectors_per_page);
}
