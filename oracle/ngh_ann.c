/* ngh_ann.c -- CPU restatement of the reference's OWN search path (SURVEY.md
 * section 8 row A11 / N3): the NGH graph index -- incremental Vamana-style insert with
 * PQ/ADC distances, beam search over PQ codes, exact re-rank of the best few.
 *
 * TEST INFRASTRUCTURE ONLY (see vs_oracle.h): it exists to put a number on what the
 * reference itself returns (recall against the exhaustive-exact answer, CPU latency) next
 * to the GPU path, which replaces this walk wholesale.  It is never linked into or called
 * from the product library.
 *
 * PARITY UNPINNED and, for this file, STATISTICAL ONLY: the reference seeds its PQ
 * training with Dart's Random(42 + m), which is not reproducible without a Dart runtime;
 * the caller supplies the seed indices, so codebooks -- and therefore graphs -- differ from
 * a real ToStore build in detail while following the same algorithm step by step.
 *
 * "ref:" = path under /root/reference/lib/src/.  Page I/O, caches, yields and the
 * same-page-only tombstone check of the beam search (ngh_graph_engine.dart:224-232) are not
 * part of the arithmetic and are left out: there are no deletes in the measurements this
 * file serves.  Everything else is restated as written, including its quirks:
 *   - the candidate min-heap silently DROPS pushes when it is full (:1158-1170)
 *   - robust prune ranks by the squared difference of PQ code BYTES (:524-532), not by a
 *     distance between centroids
 *   - inserts always search with the L2 table, whatever the index metric (:834)
 *   - cosine uses the L2 table on the un-normalised stored rows (vector_quantizer.dart:447-451)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "vs_oracle.h"

typedef struct {
  int dim, metric, M, K, sd, R, ef_search, ef_construction;
  double alpha;
  float *codebook;    /* M x K x sd */
  int64_t n, cap;     /* nodes (= nextNodeId) */
  int64_t total;      /* meta.totalVectors: grows after each batch (partition manager's vectorsDelta) */
  int64_t medoid;     /* meta.medoidNodeId */
  float *vectors;     /* n x dim (raw-vector pages) */
  uint8_t *codes;     /* n x M   (PQ-code pages) */
  uint32_t *nbr;      /* n x R   (graph slots) */
  uint8_t *deg;       /* n */
  uint32_t *stamp;    /* visited set, generation-stamped */
  uint32_t gen;
  int64_t adc_evals;  /* counters for the report */
  int64_t hops;
} vso_ann;

/* ---- _FixedHeap, ref: core/ngh_graph_engine.dart:1131-1227 ---------------------------- */
typedef struct {
  int cap, size, max_heap;
  int32_t *ids;
  double *d;
  double last_popped;
} fheap;

static void fh_init(fheap *h, int cap, int max_heap) {
  h->cap = cap;
  h->size = 0;
  h->max_heap = max_heap;
  h->ids = (int32_t *)malloc((size_t)(cap + 1) * sizeof(int32_t));
  h->d = (double *)malloc((size_t)(cap + 1) * sizeof(double));
  h->last_popped = 0;
}
static void fh_free(fheap *h) {
  free(h->ids);
  free(h->d);
}
static int fh_less(const fheap *h, int i, int j) { return h->max_heap ? h->d[i] > h->d[j] : h->d[i] < h->d[j]; }
static void fh_swap(fheap *h, int i, int j) {
  int32_t t = h->ids[i];
  double td = h->d[i];
  h->ids[i] = h->ids[j];
  h->d[i] = h->d[j];
  h->ids[j] = t;
  h->d[j] = td;
}
static void fh_up(fheap *h, int i) {
  while (i > 0) {
    int p = (i - 1) >> 1;
    if (fh_less(h, i, p)) {
      fh_swap(h, i, p);
      i = p;
    } else {
      break;
    }
  }
}
static void fh_down(fheap *h, int i) {
  for (;;) {
    int best = i, l = 2 * i + 1, r = 2 * i + 2;
    if (l < h->size && fh_less(h, l, best)) best = l;
    if (r < h->size && fh_less(h, r, best)) best = r;
    if (best == i) break;
    fh_swap(h, i, best);
    i = best;
  }
}
static int fh_full(const fheap *h) { return h->size >= h->cap; }
static double fh_peek(const fheap *h) { return h->size > 0 ? h->d[0] : INFINITY; }
static void fh_push(fheap *h, int32_t id, double dist) { /* :1158-1170 */
  if (h->size < h->cap) {
    h->ids[h->size] = id;
    h->d[h->size] = dist;
    h->size++;
    fh_up(h, h->size - 1);
  } else if (h->max_heap && dist < h->d[0]) {
    h->ids[0] = id;
    h->d[0] = dist;
    fh_down(h, 0);
  } /* a full min-heap drops the entry */
}
static int32_t fh_pop(fheap *h) { /* :1173-1183 */
  int32_t id = h->ids[0];
  h->last_popped = h->d[0];
  h->size--;
  if (h->size > 0) {
    h->ids[0] = h->ids[h->size];
    h->d[0] = h->d[h->size];
    fh_down(h, 0);
  }
  return id;
}

typedef struct {
  int32_t id;
  double d;
  int pos;
} cand_t;
static int cand_cmp(const void *a, const void *b) { /* drainSorted: compareTo on distance; ties keep heap order */
  const cand_t *x = (const cand_t *)a, *y = (const cand_t *)b;
  int c = vso_compare_double(x->d, y->d);
  if (c) return c;
  return x->pos - y->pos;
}
/* returns count; out sized h->size */
static int fh_drain_sorted(const fheap *h, cand_t *out) {
  for (int i = 0; i < h->size; ++i) {
    out[i].id = h->ids[i];
    out[i].d = h->d[i];
    out[i].pos = i;
  }
  qsort(out, (size_t)h->size, sizeof(cand_t), cand_cmp);
  return h->size;
}

/* ---- distance tables and ADC, ref: core/vector_quantizer.dart:387-458 ---------------------- */
static void build_table_l2(const vso_ann *a, const float *q, float *table) {
  for (int m = 0; m < a->M; ++m)
    for (int k = 0; k < a->K; ++k) {
      double dist = 0;
      const float *c = a->codebook + ((size_t)m * a->K + k) * a->sd;
      for (int d = 0; d < a->sd; ++d) {
        double diff = (double)q[m * a->sd + d] - (double)c[d];
        dist += diff * diff;
      }
      table[m * a->K + k] = (float)dist;
    }
}
static void build_table_ip(const vso_ann *a, const float *q, float *table) {
  for (int m = 0; m < a->M; ++m)
    for (int k = 0; k < a->K; ++k) {
      double ip = 0;
      const float *c = a->codebook + ((size_t)m * a->K + k) * a->sd;
      for (int d = 0; d < a->sd; ++d) ip += (double)q[m * a->sd + d] * (double)c[d];
      table[m * a->K + k] = (float)(-ip);
    }
}
static double adc(vso_ann *a, const float *table, const uint8_t *code) {
  double dist = 0;
  int off = 0;
  for (int m = 0; m < a->M; ++m) {
    dist += (double)table[off + code[m]];
    off += a->K;
  }
  a->adc_evals++;
  return dist;
}

/* ---- lifecycle ---------------------------------------------------------------------------- */
vso_ann *vso_ann_create(int dim, int metric, int subspaces, int centroids, int max_degree, int ef_search,
                        int ef_construction, double prune_alpha, const float *codebook) {
  vso_ann *a = (vso_ann *)calloc(1, sizeof(vso_ann));
  a->dim = dim;
  a->metric = metric;
  a->M = subspaces;
  a->K = centroids;
  a->sd = dim / subspaces;
  a->R = max_degree;
  a->ef_search = ef_search;
  a->ef_construction = ef_construction;
  a->alpha = prune_alpha;
  a->medoid = -1;
  size_t cb = (size_t)subspaces * centroids * a->sd;
  a->codebook = (float *)malloc(cb * sizeof(float));
  memcpy(a->codebook, codebook, cb * sizeof(float));
  return a;
}
void vso_ann_destroy(vso_ann *a) {
  if (!a) return;
  free(a->codebook);
  free(a->vectors);
  free(a->codes);
  free(a->nbr);
  free(a->deg);
  free(a->stamp);
  free(a);
}
static void ann_reserve(vso_ann *a, int64_t want) {
  if (want <= a->cap) return;
  int64_t nc = a->cap ? a->cap * 2 : 1024;
  while (nc < want) nc *= 2;
  a->vectors = (float *)realloc(a->vectors, (size_t)nc * a->dim * sizeof(float));
  a->codes = (uint8_t *)realloc(a->codes, (size_t)nc * a->M);
  a->nbr = (uint32_t *)realloc(a->nbr, (size_t)nc * a->R * sizeof(uint32_t));
  a->deg = (uint8_t *)realloc(a->deg, (size_t)nc);
  a->stamp = (uint32_t *)realloc(a->stamp, (size_t)nc * sizeof(uint32_t));
  memset(a->stamp + a->cap, 0, (size_t)(nc - a->cap) * sizeof(uint32_t));
  memset(a->deg + a->cap, 0, (size_t)(nc - a->cap));
  a->cap = nc;
}
static void new_generation(vso_ann *a) {
  if (++a->gen == 0) {
    memset(a->stamp, 0, (size_t)a->cap * sizeof(uint32_t));
    a->gen = 1;
  }
}

/* ---- robust prune, ref: core/ngh_graph_engine.dart:452-532 ------------------------------------- */
static double code_distance(const vso_ann *a, const uint8_t *x, const uint8_t *y) { /* :524-532 */
  double dist = 0;
  for (int i = 0; i < a->M; ++i) {
    double diff = (double)((int)x[i] - (int)y[i]);
    dist += diff * diff;
  }
  return dist;
}
typedef struct {
  int32_t id;
  double d;
  int pos;
} prune_t;
static int prune_cmp(const void *x, const void *y) {
  const prune_t *p = (const prune_t *)x, *q = (const prune_t *)y;
  int c = vso_compare_double(p->d, q->d);
  return c ? c : p->pos - q->pos;
}
/* candidates -> at most R diverse ids; returns the count */
static int robust_prune(const vso_ann *a, int64_t node, const int32_t *cands, int n_cands, int32_t *out) {
  const uint8_t *nc = a->codes + (size_t)node * a->M;
  prune_t *s = (prune_t *)malloc((size_t)n_cands * sizeof(prune_t));
  for (int i = 0; i < n_cands; ++i) {
    s[i].id = cands[i];
    s[i].d = code_distance(a, nc, a->codes + (size_t)cands[i] * a->M);
    s[i].pos = i;
  }
  qsort(s, (size_t)n_cands, sizeof(prune_t), prune_cmp);
  int cnt = 0;
  for (int i = 0; i < n_cands && cnt < a->R; ++i) {
    const uint8_t *cc = a->codes + (size_t)s[i].id * a->M;
    const double to_node = s[i].d;
    int covered = 0;
    for (int r = 0; r < cnt; ++r) {
      double cr = code_distance(a, cc, a->codes + (size_t)out[r] * a->M);
      if (a->alpha * cr < to_node) {
        covered = 1;
        break;
      }
    }
    if (!covered) out[cnt++] = s[i].id;
  }
  free(s);
  return cnt;
}

/* ---- greedy search for insert, ref: :828-902 ----------------------------------------------------- */
static int greedy_for_insert(vso_ann *a, const float *vec, int32_t *out) {
  const int ef = a->ef_construction;
  float *table = (float *)malloc((size_t)a->M * a->K * sizeof(float));
  build_table_l2(a, vec, table); /* always the L2 table */
  if (a->medoid < 0) {
    free(table);
    return 0;
  }
  fheap cand, res;
  fh_init(&cand, ef * 4, 0);
  fh_init(&res, ef, 1);
  new_generation(a);
  double ed = adc(a, table, a->codes + (size_t)a->medoid * a->M);
  a->stamp[a->medoid] = a->gen;
  fh_push(&cand, (int32_t)a->medoid, ed);
  fh_push(&res, (int32_t)a->medoid, ed);
  while (cand.size > 0) {
    int32_t cur = fh_pop(&cand);
    double cd = cand.last_popped;
    if (fh_full(&res) && cd > fh_peek(&res)) break;
    const uint32_t *nb = a->nbr + (size_t)cur * a->R;
    for (int i = 0; i < a->deg[cur]; ++i) {
      uint32_t id = nb[i];
      if (a->stamp[id] == a->gen) continue;
      a->stamp[id] = a->gen;
      double dist = adc(a, table, a->codes + (size_t)id * a->M);
      if (!fh_full(&res) || dist < fh_peek(&res)) {
        fh_push(&cand, (int32_t)id, dist);
        fh_push(&res, (int32_t)id, dist);
      }
    }
  }
  cand_t *sorted = (cand_t *)malloc((size_t)(res.size + 1) * sizeof(cand_t));
  int n = fh_drain_sorted(&res, sorted);
  for (int i = 0; i < n; ++i) out[i] = sorted[i].id;
  free(sorted);
  fh_free(&cand);
  fh_free(&res);
  free(table);
  return n;
}

/* ---- reverse edge, ref: :759-822 ------------------------------------------------------------- */
static void add_reverse_edge(vso_ann *a, int64_t neighbor, int64_t node) {
  uint32_t *nb = a->nbr + (size_t)neighbor * a->R;
  int deg = a->deg[neighbor];
  for (int j = 0; j < deg; ++j)
    if (nb[j] == (uint32_t)node) return;
  if (deg < a->R) {
    nb[deg] = (uint32_t)node;
    a->deg[neighbor] = (uint8_t)(deg + 1);
    return;
  }
  int32_t *cur = (int32_t *)malloc((size_t)(deg + 1) * sizeof(int32_t));
  int32_t *pruned = (int32_t *)malloc((size_t)a->R * sizeof(int32_t));
  for (int j = 0; j < deg; ++j) cur[j] = (int32_t)nb[j];
  cur[deg] = (int32_t)node;
  int cnt = robust_prune(a, neighbor, cur, deg + 1, pruned);
  if (cnt > a->R) cnt = a->R;
  for (int j = 0; j < cnt; ++j) nb[j] = (uint32_t)pruned[j];
  a->deg[neighbor] = (uint8_t)cnt;
  free(cur);
  free(pruned);
}

/* ---- insertBatch, ref: :297-403 (one call = one writeChanges batch) ----------------------------- */
void vso_ann_insert_batch(vso_ann *a, const float *vectors, int64_t count) {
  ann_reserve(a, a->n + count);
  int32_t *found = (int32_t *)malloc((size_t)(a->ef_construction + 1) * sizeof(int32_t));
  int32_t *pruned = (int32_t *)malloc((size_t)a->R * sizeof(int32_t));
  for (int64_t i = 0; i < count; ++i) {
    const int64_t node = a->n;
    const float *vec = vectors + (size_t)i * a->dim;
    a->n = node + 1;
    memcpy(a->vectors + (size_t)node * a->dim, vec, (size_t)a->dim * sizeof(float));
    vso_pq_encode(a->codebook, a->M, a->K, a->sd, vec, 1, a->dim, a->codes + (size_t)node * a->M); /* _quantizeVectorsBatch */
    a->deg[node] = 0;
    const int32_t *nbrs = found;
    int n_nbrs = 0;
    const int64_t existing = a->total + i; /* :332 */
    if (existing == 0) {
      a->medoid = node;
    } else if (existing < 4 || a->medoid < 0) { /* :339-351 */
      n_nbrs = (int)(existing < a->R ? existing : a->R);
      for (int j = 0; j < n_nbrs; ++j) found[j] = (int32_t)((node - existing) + j);
      if (a->medoid < 0) a->medoid = 0;
    } else {
      n_nbrs = greedy_for_insert(a, vec, found);
      if (n_nbrs > a->R) { /* :365-375 */
        n_nbrs = robust_prune(a, node, found, n_nbrs, pruned);
        nbrs = pruned;
      }
    }
    if (n_nbrs > a->R) n_nbrs = a->R; /* _writeGraphNode :744 */
    for (int j = 0; j < n_nbrs; ++j) a->nbr[(size_t)node * a->R + j] = (uint32_t)nbrs[j];
    a->deg[node] = (uint8_t)n_nbrs;
    for (int j = 0; j < n_nbrs; ++j) add_reverse_edge(a, nbrs[j], node);
  }
  a->total += count;
  free(found);
  free(pruned);
}

/* ---- search, ref: :67-135 (+ _beamSearch :145-288) -------------------------------------------- */
int64_t vso_ann_search(vso_ann *a, const float *query, int top_k, int ef_search, double threshold, int64_t *out_ids,
                       double *out_dist) {
  if (a->total == 0 || a->medoid < 0 || top_k <= 0) return 0;
  const int ef_raw = ef_search > 0 ? ef_search : a->ef_search;
  const int five = top_k * 5 > 32 ? top_k * 5 : 32;
  const int ef = ef_raw < five ? ef_raw : five; /* :83 */
  float *table = (float *)malloc((size_t)a->M * a->K * sizeof(float));
  if (a->metric == 1) build_table_ip(a, query, table);
  else build_table_l2(a, query, table); /* L2 and cosine */
  int64_t max_visited = (int64_t)ef * a->R * 3; /* :160-161 */
  if (a->total * 2 + 256 < max_visited) max_visited = a->total * 2 + 256;
  fheap cand, res;
  fh_init(&cand, ef * 4, 0);
  fh_init(&res, ef, 1);
  new_generation(a);
  int64_t visited = 1;
  double ed = adc(a, table, a->codes + (size_t)a->medoid * a->M);
  a->stamp[a->medoid] = a->gen;
  fh_push(&cand, (int32_t)a->medoid, ed);
  fh_push(&res, (int32_t)a->medoid, ed);
  while (cand.size > 0) {
    if (visited >= max_visited) break;
    int32_t cur = fh_pop(&cand);
    double cd = cand.last_popped;
    if (fh_full(&res) && cd > fh_peek(&res)) break;
    a->hops++;
    if (a->deg[cur] == 0) continue;
    const uint32_t *nb = a->nbr + (size_t)cur * a->R;
    for (int i = 0; i < a->deg[cur]; ++i) {
      uint32_t id = nb[i];
      if (a->stamp[id] == a->gen) continue;
      a->stamp[id] = a->gen;
      visited++;
      double dist = adc(a, table, a->codes + (size_t)id * a->M);
      if (!fh_full(&res) || dist < fh_peek(&res)) {
        fh_push(&cand, (int32_t)id, dist);
        fh_push(&res, (int32_t)id, dist);
      }
    }
  }
  cand_t *sorted = (cand_t *)malloc((size_t)(res.size + 1) * sizeof(cand_t));
  int n = fh_drain_sorted(&res, sorted);
  int rerank = top_k * 2 > 20 ? top_k * 2 : 20; /* :115 */
  if (rerank > n) rerank = n;
  cand_t *fin = (cand_t *)malloc((size_t)(rerank + 1) * sizeof(cand_t));
  int m = 0;
  for (int i = 0; i < rerank; ++i) {
    double ex = vso_exact_distance(query, a->vectors + (size_t)sorted[i].id * a->dim, a->dim, a->metric);
    if (!isnan(threshold) && ex > threshold) continue;
    fin[m].id = sorted[i].id;
    fin[m].d = ex;
    fin[m].pos = m;
    m++;
  }
  qsort(fin, (size_t)m, sizeof(cand_t), cand_cmp);
  if (m > top_k) m = top_k;
  for (int i = 0; i < m; ++i) {
    out_ids[i] = fin[i].id;
    out_dist[i] = fin[i].d;
  }
  free(fin);
  free(sorted);
  fh_free(&cand);
  fh_free(&res);
  free(table);
  return m;
}

int64_t vso_ann_size(const vso_ann *a) { return a->n; }
void vso_ann_counters(vso_ann *a, int64_t *adc_evals, int64_t *hops, int reset) {
  if (adc_evals) *adc_evals = a->adc_evals;
  if (hops) *hops = a->hops;
  if (reset) a->adc_evals = a->hops = 0;
}
/* mean out-degree, for the report */
double vso_ann_mean_degree(const vso_ann *a) {
  double s = 0;
  for (int64_t i = 0; i < a->n; ++i) s += a->deg[i];
  return a->n ? s / (double)a->n : 0;
}
