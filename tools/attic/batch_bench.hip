// batch_bench.hip -- correctness + throughput harness for tsh::batch_score_kernel
// (not part of the product library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o batch_bench batch_bench.hip
//   ./batch_bench [n=1000000] [d=768] [nq=1024] [iters=3]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../tostore_amd/csrc/tsh_batch.hip.h"

using namespace tsh;

#define CK(x)                                                          \
  do {                                                                 \
    hipError_t e_ = (x);                                               \
    if (e_ != hipSuccess) {                                            \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));          \
      exit(1);                                                         \
    }                                                                  \
  } while (0)

static uint32_t rng_state = 12345u;
static float frand() {
  rng_state = rng_state * 1664525u + 1013904223u;
  return ((rng_state >> 8) & 0xFFFF) / 32768.0f - 1.0f;
}

template <int METRIC>
int check(int n, int d, int nq) {
  int ld = (d + 3) / 4 * 4, nq_pad = (nq + BT_M - 1) / BT_M * BT_M;
  std::vector<float> V((size_t)n * ld, 0.f), Q((size_t)nq_pad * ld, 0.f), inv(n), sq(n), qsq(nq_pad, 0.f);
  for (int i = 0; i < n; ++i) {
    double s = 0;
    for (int k = 0; k < d; ++k) { float x = frand(); V[(size_t)i * ld + k] = x; s += (double)x * x; }
    inv[i] = (float)(1.0 / std::sqrt(s)); sq[i] = (float)s;
  }
  for (int i = 0; i < nq; ++i) {
    double s = 0;
    for (int k = 0; k < d; ++k) { float x = frand(); Q[(size_t)i * ld + k] = x; s += (double)x * x; }
    qsq[i] = (float)s;
  }
  float *dV, *dQ, *dinv, *dsq, *dqsq, *dthr, *dd;
  uint32_t *ck, *cr, *cc;
  int cap = 4096;
  CK(hipMalloc(&dV, V.size() * 4)); CK(hipMalloc(&dQ, Q.size() * 4)); CK(hipMalloc(&dinv, n * 4));
  CK(hipMalloc(&dsq, n * 4)); CK(hipMalloc(&dqsq, nq_pad * 4)); CK(hipMalloc(&dthr, nq_pad * 4));
  CK(hipMalloc(&dd, (size_t)nq_pad * n * 4));
  CK(hipMalloc(&ck, (size_t)nq * cap * 4)); CK(hipMalloc(&cr, (size_t)nq * cap * 4)); CK(hipMalloc(&cc, nq * 4));
  CK(hipMemcpy(dV, V.data(), V.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dQ, Q.data(), Q.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dinv, inv.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsq, sq.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dqsq, qsq.data(), nq_pad * 4, hipMemcpyHostToDevice));
  BatchArgs a{};
  a.Q = dQ; a.V = dV; a.inv_norm = dinv; a.sqnorm = dsq; a.qsq = dqsq; a.thr = dthr; a.dense = dd;
  a.cand_key = ck; a.cand_row = cr; a.cand_cnt = cc; a.ld = ld; a.dense_ld = n; a.row0 = 0; a.row1 = n;
  a.nq = nq; a.nq_pad = nq_pad; a.kchunks = (ld + BT_K - 1) / BT_K; a.cand_cap = cap;
  a.q_tiles = nq_pad / BT_M; a.n_tiles = (n + BT_N - 1) / BT_N;
  batch_score_kernel<METRIC, true><<<a.q_tiles * a.n_tiles, BT_THREADS>>>(a);
  CK(hipDeviceSynchronize());
  std::vector<float> D((size_t)nq_pad * n);
  CK(hipMemcpy(D.data(), dd, D.size() * 4, hipMemcpyDeviceToHost));
  double maxerr = 0;
  for (int q = 0; q < nq; ++q)
    for (int i = 0; i < n; ++i) {
      double dot = 0;
      for (int k = 0; k < d; ++k) dot += (double)Q[(size_t)q * ld + k] * V[(size_t)i * ld + k];
      double key = METRIC == METRIC_IP ? -dot : METRIC == METRIC_COS ? -dot * inv[i] : (double)qsq[q] + sq[i] - 2 * dot;
      double e = std::fabs(key - D[(size_t)q * n + i]) / (1.0 + std::fabs(key));
      if (e > maxerr) maxerr = e;
    }
  // filter mode must select exactly the keys <= thr of the dense output
  std::vector<float> thr(nq_pad, -1e30f);
  for (int q = 0; q < nq; ++q) {  // threshold = 20th smallest dense key of that query
    std::vector<float> row(D.begin() + (size_t)q * n, D.begin() + (size_t)q * n + n);
    std::nth_element(row.begin(), row.begin() + 19, row.end());
    thr[q] = row[19];
  }
  CK(hipMemcpy(dthr, thr.data(), nq_pad * 4, hipMemcpyHostToDevice));
  CK(hipMemset(cc, 0, nq * 4));
  batch_score_kernel<METRIC, false><<<a.q_tiles * a.n_tiles, BT_THREADS>>>(a);
  CK(hipDeviceSynchronize());
  std::vector<uint32_t> cnt(nq), rows((size_t)nq * cap), keys((size_t)nq * cap);
  CK(hipMemcpy(cnt.data(), cc, nq * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(rows.data(), cr, rows.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(keys.data(), ck, keys.size() * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int q = 0; q < nq; ++q) {
    int want = 0;
    for (int i = 0; i < n; ++i) want += D[(size_t)q * n + i] <= thr[q];
    if ((int)cnt[q] != want) ++bad;
    for (uint32_t c = 0; c < cnt[q] && c < (uint32_t)cap; ++c) {
      float kf; uint32_t kb = keys[(size_t)q * cap + c]; memcpy(&kf, &kb, 4);
      if (kf != D[(size_t)q * n + rows[(size_t)q * cap + c]]) ++bad;
    }
  }
  printf("metric %d  n=%d d=%d nq=%d: dense max rel err %.3g  filter mismatches %d  %s\n", METRIC, n, d, nq, maxerr, bad,
         (maxerr < 2e-5 && bad == 0) ? "OK" : "FAIL");
  hipFree(dV); hipFree(dQ); hipFree(dinv); hipFree(dsq); hipFree(dqsq); hipFree(dthr); hipFree(dd);
  hipFree(ck); hipFree(cr); hipFree(cc);
  return (maxerr < 2e-5 && bad == 0) ? 0 : 1;
}

// unit-variance-ish normal data scaled to |row| ~ 1 for d = 768 (Box-Muller on a hash)
__global__ void fill_normal_kernel(float *p, size_t n, uint32_t seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed, y = (uint32_t)(i >> 7) * 40503u + seed * 977u + (uint32_t)i;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    y ^= y >> 15; y *= 0x2c1b3c6du; y ^= y >> 12; y *= 0x297a2d39u; y ^= y >> 15;
    float u1 = ((x >> 8) + 1) / 16777217.0f, u2 = (y >> 8) / 16777216.0f;
    p[i] = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2) * scale;
  }
}

__global__ void fill_kernel(float *p, size_t n, uint32_t seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = ((float)(x & 0xFFFFFF) / 8388608.0f - 1.0f) * 0.05f;
  }
}

int main(int argc, char **argv) {
  int n = argc > 1 ? atoi(argv[1]) : 1000000, d = argc > 2 ? atoi(argv[2]) : 768;
  int nq = argc > 3 ? atoi(argv[3]) : 1024, iters = argc > 4 ? atoi(argv[4]) : 3;
  int fails = 0;
  fails += check<METRIC_IP>(1000, 100, 200);
  fails += check<METRIC_COS>(777, 768, 130);
  fails += check<METRIC_L2>(1290, 36, 256);
  fails += check<METRIC_COS>(300, 7, 5);
  int ld = (d + 3) / 4 * 4, nq_pad = (nq + BT_M - 1) / BT_M * BT_M;
  float *dV, *dQ, *dinv, *dthr;
  uint32_t *ck, *cr, *cc;
  CK(hipMalloc(&dV, (size_t)n * ld * 4)); CK(hipMalloc(&dQ, (size_t)nq_pad * ld * 4));
  CK(hipMalloc(&dinv, (size_t)n * 4)); CK(hipMalloc(&dthr, nq_pad * 4));
  CK(hipMalloc(&ck, (size_t)nq * 1024 * 4)); CK(hipMalloc(&cr, (size_t)nq * 1024 * 4)); CK(hipMalloc(&cc, nq * 4));
  fill_kernel<<<4096, 256>>>(dV, (size_t)n * ld, 1u);
  fill_kernel<<<64, 256>>>(dQ, (size_t)nq_pad * ld, 2u);
  fill_kernel<<<64, 256>>>(dinv, n, 3u);
  std::vector<float> thr(nq_pad, -1e30f);
  CK(hipMemcpy(dthr, thr.data(), nq_pad * 4, hipMemcpyHostToDevice));
  CK(hipMemset(cc, 0, nq * 4));
  BatchArgs a{};
  a.Q = dQ; a.V = dV; a.inv_norm = dinv; a.thr = dthr; a.cand_key = ck; a.cand_row = cr; a.cand_cnt = cc;
  a.ld = ld; a.row0 = 0; a.row1 = n; a.nq = nq; a.nq_pad = nq_pad; a.kchunks = (ld + BT_K - 1) / BT_K;
  a.cand_cap = 1024; a.q_tiles = nq_pad / BT_M; a.n_tiles = (n + BT_N - 1) / BT_N;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  batch_score_kernel<METRIC_COS, false><<<a.q_tiles * a.n_tiles, BT_THREADS>>>(a);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) batch_score_kernel<METRIC_COS, false><<<a.q_tiles * a.n_tiles, BT_THREADS>>>(a);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= iters;
  double flop = 2.0 * nq * (double)n * d;
  printf("batch_score_kernel<COS,filter,BK=32> nq=%d n=%d d=%d: %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)  %.0f queries/s\n", nq, n, d,
         ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100, nq / (ms * 1e-3));
  // realistic data: N(0,1/d) rows and queries, inv_norm = 1, threshold at ~2.8 sigma (0.26 % pass)
  fill_normal_kernel<<<4096, 256>>>(dV, (size_t)n * ld, 11u, 1.0f / sqrtf((float)d));
  fill_normal_kernel<<<64, 256>>>(dQ, (size_t)nq_pad * ld, 12u, 1.0f / sqrtf((float)d));
  { std::vector<float> ones(n, 1.0f); CK(hipMemcpy(dinv, ones.data(), (size_t)n * 4, hipMemcpyHostToDevice)); }
  for (float t : {-1e30f, -0.10f}) {
    std::vector<float> th(nq_pad, t);
    CK(hipMemcpy(dthr, th.data(), nq_pad * 4, hipMemcpyHostToDevice));
    CK(hipMemset(cc, 0, nq * 4));
    a.kchunks = (ld + 31) / 32;
    batch_score_kernel<METRIC_COS, false, 32><<<a.q_tiles * a.n_tiles, BT_THREADS>>>(a);
    CK(hipMemset(cc, 0, nq * 4));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) batch_score_kernel<METRIC_COS, false, 32><<<a.q_tiles * a.n_tiles, BT_THREADS>>>(a);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    std::vector<uint32_t> hc(nq);
    CK(hipMemcpy(hc.data(), cc, nq * 4, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : hc) avg += v; avg /= nq * (double)iters;
    printf("normal data, thr=%g: %.3f ms  %.1f TFLOP/s (%.1f%%)  survivors/query %.0f\n", t, ms, flop / ms / 1e9,
           flop / ms / 1e9 / 157.3 * 100, avg);
  }
  a.kchunks = (ld + 15) / 16;
  batch_score_kernel<METRIC_COS, false, 16><<<a.q_tiles * a.n_tiles, BT_THREADS>>>(a);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) batch_score_kernel<METRIC_COS, false, 16><<<a.q_tiles * a.n_tiles, BT_THREADS>>>(a);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= iters;
  printf("batch_score_kernel<COS,filter,BK=16> nq=%d n=%d d=%d: %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)  %.0f queries/s\n", nq, n, d,
         ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100, nq / (ms * 1e-3));
  return fails;
}
