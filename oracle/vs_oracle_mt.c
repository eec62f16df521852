/*
 * vs_oracle_mt.c -- OpenMP wrapper around the scalar restatement: the same
 * per-row arithmetic (vso_exact_distance), rows split into contiguous ranges,
 * one bounded heap per thread, merged and sorted at the end.
 * TEST INFRASTRUCTURE ONLY: the "fair" CPU baseline of BASELINE.md section 3.
 */
#include "vs_oracle.h"

#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

int vso_mt_max_threads(void) { return omp_get_max_threads(); }

int64_t vso_search_heap_mt(const float *rows, int64_t n, int d, int metric,
                           const float *query, int64_t k, double threshold,
                           const uint8_t *keep, int threads, int64_t *out_ids,
                           double *out_dist) {
  int T = threads > 0 ? threads : omp_get_max_threads();
  int64_t *ids;
  double *dist;
  int64_t *cnt;
  int64_t total = 0, i, r;
  if (n <= 0 || k <= 0) return 0;
  ids = (int64_t *)malloc((size_t)T * (size_t)k * sizeof(int64_t));
  dist = (double *)malloc((size_t)T * (size_t)k * sizeof(double));
  cnt = (int64_t *)calloc((size_t)T, sizeof(int64_t));
  if (!ids || !dist || !cnt) return -1;
#pragma omp parallel num_threads(T)
  {
    int t = omp_get_thread_num();
    /* 8-row aligned split so a byte of `keep` never straddles two threads */
    int64_t chunk = ((n + T - 1) / T + 7) & ~(int64_t)7;
    int64_t lo = (int64_t)t * chunk, hi = lo + chunk;
    if (hi > n) hi = n;
    if (lo < hi) {
      int64_t m = vso_search_heap(rows + lo * (int64_t)d, hi - lo, d, metric, query, k,
                                  threshold, keep ? keep + (lo >> 3) : NULL,
                                  ids + (int64_t)t * k, dist + (int64_t)t * k);
      int64_t j;
      for (j = 0; j < m; j++) ids[(int64_t)t * k + j] += lo;
      cnt[t] = m;
    }
  }
  /* merge: gather, then reuse the exhaustive sorter on (dist,id) pairs */
  {
    typedef struct { double dist; int64_t id; } hit;
    hit *all;
    int t;
    for (t = 0; t < T; t++) total += cnt[t];
    all = (hit *)malloc((size_t)(total ? total : 1) * sizeof(hit));
    total = 0;
    for (t = 0; t < T; t++)
      for (i = 0; i < cnt[t]; i++) {
        all[total].dist = dist[(int64_t)t * k + i];
        all[total].id = ids[(int64_t)t * k + i];
        total++;
      }
    /* insertion sort is enough: total <= T*k */
    for (i = 1; i < total; i++) {
      hit h = all[i];
      int64_t j = i - 1;
      while (j >= 0) {
        int c = vso_compare_double(all[j].dist, h.dist);
        if (c < 0 || (c == 0 && all[j].id < h.id)) break;
        all[j + 1] = all[j];
        j--;
      }
      all[j + 1] = h;
    }
    r = total < k ? total : k;
    for (i = 0; i < r; i++) {
      out_ids[i] = all[i].id;
      out_dist[i] = all[i].dist;
    }
    free(all);
  }
  free(ids);
  free(dist);
  free(cnt);
  return r;
}

/* Many queries at once (recall checks over >= 1000 queries): OpenMP over groups of
 * QG queries; within a group every row is visited once and the QG distances are
 * QG independent accumulation chains, each in the reference's element order, so
 * every (query,row) distance is bit-identical to vso_exact_distance.  Results:
 * out_ids / out_dist are nq x k (row-major), out_count nq. */
#define QG 4
typedef struct { double dist; int64_t id; } mhit;

static int mhit_cmp(const mhit *a, const mhit *b) {
  int c = vso_compare_double(a->dist, b->dist);
  if (c) return c;
  return a->id < b->id ? -1 : (a->id > b->id ? 1 : 0);
}
static void msift_down(mhit *h, int64_t n, int64_t i) {
  for (;;) {
    int64_t l = 2 * i + 1, r = l + 1, m = i;
    mhit t;
    if (l < n && mhit_cmp(&h[l], &h[m]) > 0) m = l;
    if (r < n && mhit_cmp(&h[r], &h[m]) > 0) m = r;
    if (m == i) return;
    t = h[i]; h[i] = h[m]; h[m] = t;
    i = m;
  }
}
static int mhit_qsort(const void *a, const void *b) { return mhit_cmp((const mhit *)a, (const mhit *)b); }

int64_t vso_search_heap_many_mt(const float *rows, int64_t n, int d, int metric, const float *queries,
                                int64_t nq, int64_t k, double threshold, const uint8_t *keep, int threads,
                                int64_t *out_ids, double *out_dist, int64_t *out_count) {
  int T = threads > 0 ? threads : omp_get_max_threads();
  int64_t groups = (nq + QG - 1) / QG, g;
  int failed = 0;
  if (k <= 0) {
    for (g = 0; g < nq; g++) out_count[g] = 0;
    return 0;
  }
#pragma omp parallel for num_threads(T) schedule(dynamic, 1)
  for (g = 0; g < groups; g++) {
    int64_t q0 = g * QG, nqg = nq - q0 < QG ? nq - q0 : QG, i;
    mhit *heap = (mhit *)malloc((size_t)QG * (size_t)k * sizeof(mhit));
    int64_t m[QG] = {0, 0, 0, 0};
    int u;
    if (!heap) { failed = 1; continue; }
    for (i = 0; i < n; i++) {
      const float *b = rows + i * (int64_t)d;
      double s0[QG] = {0, 0, 0, 0}, s1[QG] = {0, 0, 0, 0}, s2[QG] = {0, 0, 0, 0};
      int e;
      if (keep && !((keep[i >> 3] >> (i & 7)) & 1)) continue;
      for (e = 0; e < d; e++) {
        double bv = (double)b[e];
        for (u = 0; u < QG; u++) {  /* QG independent chains; unused ones read query 0 again */
          double av = (double)queries[(q0 + (u < nqg ? u : 0)) * (int64_t)d + e];
          if (metric == VSO_L2) {
            double diff = av - bv;
            s0[u] += diff * diff;
          } else {
            s0[u] += av * bv;
            if (metric == VSO_COSINE) { s1[u] += av * av; s2[u] += bv * bv; }
          }
        }
      }
      for (u = 0; u < nqg; u++) {
        mhit h;
        mhit *hp = heap + (int64_t)u * k;
        if (metric == VSO_L2) h.dist = sqrt(s0[u]);
        else if (metric == VSO_IP) h.dist = -s0[u];
        else { double den = sqrt(s1[u]) * sqrt(s2[u]); h.dist = 1.0 - (den > 0 ? s0[u] / den : 0); }
        h.id = i;
        if (!isnan(threshold) && h.dist > threshold) continue;
        if (m[u] < k) {
          int64_t c = m[u]++;
          hp[c] = h;
          while (c > 0) {
            int64_t p = (c - 1) / 2;
            mhit t;
            if (mhit_cmp(&hp[c], &hp[p]) <= 0) break;
            t = hp[c]; hp[c] = hp[p]; hp[p] = t;
            c = p;
          }
        } else if (mhit_cmp(&h, &hp[0]) < 0) {
          hp[0] = h;
          msift_down(hp, m[u], 0);
        }
      }
    }
    for (u = 0; u < nqg; u++) {
      mhit *hp = heap + (int64_t)u * k;
      int64_t j;
      qsort(hp, (size_t)m[u], sizeof(mhit), mhit_qsort);
      for (j = 0; j < m[u]; j++) {
        out_ids[(q0 + u) * k + j] = hp[j].id;
        out_dist[(q0 + u) * k + j] = hp[j].dist;
      }
      out_count[q0 + u] = m[u];
    }
    free(heap);
  }
  return failed ? -1 : 0;
}
