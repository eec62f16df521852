/* cabi_driver.c -- a host program with nothing but the C ABI (no Python, no torch): the situation of the
 * Dart process that binds libtostore_hip.so through dart:ffi.
 *   gcc -std=c99 -O2 -I include tools/cabi_driver.c -o tools/cabi_driver -L tostore_amd -ltostore_hip \
 *       -Wl,-rpath,$PWD/tostore_amd -lm
 *   ./tools/cabi_driver [rows=200000] [dim=768] [queries=256] [k=100]
 * Appends pseudo-random rows, checks that every row finds itself first (distance 0 under L2), times single
 * queries one at a time, a pipelined multi-query call and a batched call. */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "tostore_hip.h"

static double now(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
static unsigned int st = 12345u;
static float frand(void) {
  st = st * 1664525u + 1013904223u;
  return (float)(st >> 16) / 32768.0f - 1.0f; /* high bits: the low bits of an LCG repeat early */
}
#define CHECK(x)                                                     \
  do {                                                               \
    int rc_ = (x);                                                   \
    if (rc_ != TSH_OK) {                                             \
      char b_[256];                                                  \
      tsh_last_error(b_, (int32_t)sizeof b_);                        \
      fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, b_);          \
      return 1;                                                      \
    }                                                                \
  } while (0)

int main(int argc, char **argv) {
  int64_t n = argc > 1 ? atoll(argv[1]) : 200000;
  int d = argc > 2 ? atoi(argv[2]) : 768, nq = argc > 3 ? atoi(argv[3]) : 256, k = argc > 4 ? atoi(argv[4]) : 100;
  printf("abi %d, devices %d\n", tsh_abi_version(), tsh_device_count());
  float *rows = (float *)malloc((size_t)n * d * sizeof(float));
  for (int64_t i = 0; i < n * d; ++i) rows[i] = frand();
  tsh_index *idx = NULL;
  CHECK(tsh_index_create(d, TSH_METRIC_L2, n, 1, &idx));
  double t = now();
  CHECK(tsh_index_append(idx, 0, n, rows));
  printf("append %lld x %d: %.1f ms\n", (long long)n, d, (now() - t) * 1e3);
  int64_t *ids = (int64_t *)malloc((size_t)nq * k * sizeof(int64_t));
  double *dist = (double *)malloc((size_t)nq * k * sizeof(double));
  int32_t *cnt = (int32_t *)malloc((size_t)nq * sizeof(int32_t));
  const float *qs = rows + (size_t)(n / 3) * d; /* nq consecutive stored rows as queries */
  CHECK(tsh_search(idx, qs, 1, k, NAN, NULL, ids, dist, cnt)); /* warm-up */
  t = now();
  for (int q = 0; q < nq; ++q) CHECK(tsh_search(idx, qs + (size_t)q * d, 1, k, NAN, NULL, ids, dist, cnt));
  double one = (now() - t) / nq;
  CHECK(tsh_index_set_option(idx, TSH_OPT_BATCH_MIN_NQ, 0));
  t = now();
  CHECK(tsh_search(idx, qs, nq, k, NAN, NULL, ids, dist, cnt));
  double piped = (now() - t) / nq;
  int bad = 0;
  for (int q = 0; q < nq; ++q)
    if (cnt[q] != k || ids[(size_t)q * k] != n / 3 + q || dist[(size_t)q * k] != 0.0) {
      if (!bad) fprintf(stderr, "query %d: count %d, first id %lld (want %lld), first distance %g\n", q, cnt[q],
                        (long long)ids[(size_t)q * k], (long long)(n / 3 + q), dist[(size_t)q * k]);
      ++bad;
    }
  CHECK(tsh_index_set_option(idx, TSH_OPT_BATCH_MIN_NQ, 2));
  CHECK(tsh_search(idx, qs, nq, k, NAN, NULL, ids, dist, cnt)); /* builds the converted rows */
  t = now();
  CHECK(tsh_search(idx, qs, nq, k, NAN, NULL, ids, dist, cnt));
  double batched = (now() - t) / nq;
  for (int q = 0; q < nq; ++q)
    if (cnt[q] != k || ids[(size_t)q * k] != n / 3 + q || dist[(size_t)q * k] != 0.0) ++bad;
  tsh_counters c;
  CHECK(tsh_get_counters(idx, &c));
  printf("one at a time %.1f us/query, pipelined %.1f us/query, batched %.2f us/query; self-hits wrong: %d; "
         "scans %lld, batches %lld, fallbacks %lld\n", one * 1e6, piped * 1e6, batched * 1e6, bad,
         (long long)c.scan_launches, (long long)c.batch_launches, (long long)c.fallback_searches);
  CHECK(tsh_index_destroy(idx));
  puts(bad ? "FAILED" : "ok");
  return bad != 0;
}
