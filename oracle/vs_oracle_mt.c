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
