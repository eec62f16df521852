// batch_bf16_bench.hip -- correctness + throughput harness for tsh::batch_score_bf16x3_kernel
// (not part of the product library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o batch_bf16_bench batch_bf16_bench.hip
//   ./batch_bf16_bench [n=1000000] [d=768] [nq=1024] [iters=3]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../tostore_amd/csrc/tsh_batch.hip.h"

using namespace tsh;

#define CK(x)                                                 \
  do {                                                        \
    hipError_t e_ = (x);                                      \
    if (e_ != hipSuccess) {                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
      exit(1);                                                \
    }                                                         \
  } while (0)

static uint32_t rng_state = 777u;
static float frand() {
  rng_state = rng_state * 1664525u + 1013904223u;
  return ((rng_state >> 8) & 0xFFFF) / 32768.0f - 1.0f;
}

static void split(const float *src, int64_t ld, int64_t n, int dim, int hch, u32x4 *out) {
  SplitArgs sa{};
  sa.rows = src; sa.out = out; sa.ld = ld; sa.first = 0; sa.n = n; sa.dim = dim; sa.hchunks = hch;
  int64_t total = n * hch * 4;
  split_rows_kernel<<<(int)std::min<int64_t>((total + 255) / 256, 65536), 256>>>(sa);
}

template <int METRIC, int TM = 128, int TN = 128, int PM = 64, int KIND = 0>
int check(int n, int d, int nq, float scale) {
  int ld = (d + 3) / 4 * 4, nq_pad = (nq + TM - 1) / TM * TM, hch = (d + 31) / 32;
  constexpr int THREADS = HbTile<TM, TN, PM>::THREADS;
  std::vector<float> V((size_t)n * ld, 0.f), Q((size_t)nq_pad * ld, 0.f), inv(n), sq(n), qsq(nq_pad, 0.f);
  double vmax = 0, qmax = 0;
  for (int i = 0; i < n; ++i) {
    double s = 0;
    for (int k = 0; k < d; ++k) { float x = frand() * scale; V[(size_t)i * ld + k] = x; s += (double)x * x; }
    inv[i] = (float)(1.0 / std::sqrt(s)); sq[i] = (float)s; vmax = std::max(vmax, std::sqrt(s));
  }
  for (int i = 0; i < nq; ++i) {
    double s = 0;
    for (int k = 0; k < d; ++k) { float x = frand(); Q[(size_t)i * ld + k] = x; s += (double)x * x; }
    qsq[i] = (float)s; qmax = std::max(qmax, std::sqrt(s));
  }
  float *dV, *dQ, *dinv, *dsq, *dqsq, *dthr, *dd;
  u32x4 *dVs, *dQs;
  uint32_t *ck, *cr, *cc;
  int cap = 4096;
  CK(hipMalloc(&dV, V.size() * 4)); CK(hipMalloc(&dQ, Q.size() * 4)); CK(hipMalloc(&dinv, n * 4));
  CK(hipMalloc(&dsq, n * 4)); CK(hipMalloc(&dqsq, nq_pad * 4)); CK(hipMalloc(&dthr, nq_pad * 4));
  CK(hipMalloc(&dd, (size_t)nq_pad * n * 4));
  CK(hipMalloc(&dVs, (size_t)((n + 255) / 256 * 256) * hch * 128)); CK(hipMalloc(&dQs, (size_t)((nq_pad + 255) / 256 * 256) * hch * 128));
  CK(hipMalloc(&ck, (size_t)nq * cap * 4)); CK(hipMalloc(&cr, (size_t)nq * cap * 4)); CK(hipMalloc(&cc, nq * 4));
  CK(hipMemcpy(dV, V.data(), V.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dQ, Q.data(), Q.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dinv, inv.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsq, sq.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dqsq, qsq.data(), nq_pad * 4, hipMemcpyHostToDevice));
  split(dV, ld, n, d, hch, dVs);
  split(dQ, ld, nq_pad, d, hch, dQs);
  BatchArgs a{};
  a.Q = dQ; a.V = dV; a.Qs = dQs; a.Vs = dVs; a.hchunks = hch;
  a.inv_norm = dinv; a.sqnorm = dsq; a.qsq = dqsq; a.thr = dthr; a.dense = dd;
  a.cand_key = ck; a.cand_row = cr; a.cand_cnt = cc; a.ld = ld; a.dense_ld = n; a.row0 = 0; a.row1 = n;
  a.nq = nq; a.nq_pad = nq_pad; a.kchunks = (ld + BT_K - 1) / BT_K; a.cand_cap = cap;
  a.q_tiles = nq_pad / TM; a.n_tiles = (n + TN - 1) / TN;
  batch_score_bf16x3_kernel<METRIC, true, TM, TN, PM><<<a.q_tiles * a.n_tiles, THREADS>>>(a);
  CK(hipDeviceSynchronize());
  std::vector<float> D((size_t)nq_pad * n);
  CK(hipMemcpy(D.data(), dd, D.size() * 4, hipMemcpyDeviceToHost));
  // error of the DOT relative to |q||v| (the quantity the band is derived from)
  double maxerr = 0;
  for (int q = 0; q < nq; ++q)
    for (int i = 0; i < n; ++i) {
      double dot = 0;
      for (int k = 0; k < d; ++k) dot += (double)Q[(size_t)q * ld + k] * V[(size_t)i * ld + k];
      double got = D[(size_t)q * n + i], gdot;
      if (METRIC == METRIC_IP) gdot = -got;
      else if (METRIC == METRIC_COS) gdot = -got / inv[i];
      else gdot = ((double)qsq[q] + sq[i] - got) / 2;
      double e = std::fabs(gdot - dot) / (std::sqrt((double)qsq[q]) * std::sqrt((double)sq[i]) + 1e-300);
      if (METRIC == METRIC_L2) {  // the key adds |q|^2 + |v|^2 in f32: allow its 6 u (|q|^2 + |v|^2) / (2 |q||v|)
        double slack = 6.0 / 8388608.0 * ((double)qsq[q] + sq[i]) / (2 * std::sqrt((double)qsq[q]) * std::sqrt((double)sq[i]) + 1e-300);
        e = e > slack ? e - slack : 0;
      }
      if (e > maxerr) maxerr = e;
    }
  std::vector<float> thr(nq_pad, -1e30f);
  for (int q = 0; q < nq; ++q) {
    std::vector<float> row(D.begin() + (size_t)q * n, D.begin() + (size_t)q * n + n);
    std::nth_element(row.begin(), row.begin() + 19, row.end());
    thr[q] = row[19];
  }
  CK(hipMemcpy(dthr, thr.data(), nq_pad * 4, hipMemcpyHostToDevice));
  CK(hipMemset(cc, 0, nq * 4));
  batch_score_bf16x3_kernel<METRIC, false, TM, TN, PM><<<a.q_tiles * a.n_tiles, THREADS>>>(a);
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
  // bound: 3.1 * 2^-18 (representation) + (3 ld + 8) 2^-23 (accumulation, worst case)
  double bound = 3.1 / 262144.0 + (3.0 * ld + 8) / 8388608.0;
  bool ok = maxerr < bound && bad == 0;
  printf("kind %d tile %dx%d metric %d  n=%d d=%d nq=%d scale=%g: dot err / |q||v| max %.3g (bound %.3g)  filter mismatches %d  %s\n", KIND, TM, TN, METRIC, n,
         d, nq, scale, maxerr, bound, bad, ok ? "OK" : "FAIL");
  hipFree(dV); hipFree(dQ); hipFree(dinv); hipFree(dsq); hipFree(dqsq); hipFree(dthr); hipFree(dd);
  hipFree(dVs); hipFree(dQs); hipFree(ck); hipFree(cr); hipFree(cc);
  return ok ? 0 : 1;
}

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

int main(int argc, char **argv) {
  int n = argc > 1 ? atoi(argv[1]) : 1000000, d = argc > 2 ? atoi(argv[2]) : 768;
  int nq = argc > 3 ? atoi(argv[3]) : 1024, iters = argc > 4 ? atoi(argv[4]) : 3;
  int fails = 0;
  fails += check<METRIC_IP>(1000, 100, 200, 1.f);
  fails += check<METRIC_COS>(777, 768, 130, 1.f);
  fails += check<METRIC_L2>(1290, 36, 256, 1.f);
  fails += check<METRIC_COS>(300, 7, 5, 1.f);
  fails += check<METRIC_IP>(513, 1536, 129, 1e6f);
  fails += check<METRIC_L2>(400, 96, 64, 1e-6f);
  fails += (check<METRIC_IP, 256, 256, 128>(1000, 100, 200, 1.f));
  fails += (check<METRIC_COS, 256, 256, 128>(777, 768, 300, 1.f));
  fails += (check<METRIC_L2, 256, 256, 128>(1290, 36, 256, 1.f));
  fails += (check<METRIC_COS, 256, 256, 128>(300, 7, 5, 1.f));
  int ld = (d + 3) / 4 * 4, nq_pad = (nq + 255) / 256 * 256, hch = (d + 31) / 32;
  float *dV, *dQ, *dinv, *dthr;
  u32x4 *dVs, *dQs;
  uint32_t *ck, *cr, *cc;
  CK(hipMalloc(&dV, (size_t)n * ld * 4)); CK(hipMalloc(&dQ, (size_t)nq_pad * ld * 4));
  CK(hipMalloc(&dVs, (size_t)((n + 255) / 256 * 256) * hch * 128)); CK(hipMalloc(&dQs, (size_t)((nq_pad + 255) / 256 * 256) * hch * 128));
  CK(hipMalloc(&dinv, (size_t)n * 4)); CK(hipMalloc(&dthr, nq_pad * 4));
  CK(hipMalloc(&ck, (size_t)nq * 1024 * 4)); CK(hipMalloc(&cr, (size_t)nq * 1024 * 4)); CK(hipMalloc(&cc, nq * 4));
  fill_normal_kernel<<<4096, 256>>>(dV, (size_t)n * ld, 11u, 1.0f / sqrtf((float)d));
  fill_normal_kernel<<<64, 256>>>(dQ, (size_t)nq_pad * ld, 12u, 1.0f / sqrtf((float)d));
  { std::vector<float> ones(n, 1.0f); CK(hipMemcpy(dinv, ones.data(), (size_t)n * 4, hipMemcpyHostToDevice)); }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  CK(hipEventRecord(e0, 0));
  split(dV, ld, n, d, hch, dVs);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("split_rows_kernel %d x %d: %.3f ms (%.0f GB/s read+write)\n", n, d, ms, 2.0 * n * hch * 128 / ms / 1e6);
  split(dQ, ld, nq_pad, d, hch, dQs);
  BatchArgs a{};
  a.Q = dQ; a.V = dV; a.Qs = dQs; a.Vs = dVs; a.hchunks = hch; a.inv_norm = dinv; a.thr = dthr;
  a.cand_key = ck; a.cand_row = cr; a.cand_cnt = cc;
  a.ld = ld; a.row0 = 0; a.row1 = n; a.nq = nq; a.nq_pad = nq_pad; a.kchunks = (ld + BT_K - 1) / BT_K;
  a.cand_cap = 1024; a.q_tiles = nq_pad / BT_M; a.n_tiles = (n + BT_N - 1) / BT_N;
  double flop = 2.0 * nq * (double)n * d;
  auto time_tile = [&](auto TMc, auto TNc, auto PMc) {
    constexpr int TM = decltype(TMc)::value, TN = decltype(TNc)::value, PM = decltype(PMc)::value;
    constexpr int THREADS = HbTile<TM, TN, PM>::THREADS;
    BatchArgs b = a;
    b.q_tiles = nq_pad / TM;
    b.n_tiles = (n + TN - 1) / TN;
    for (float t : {-1e30f, -0.10f}) {
      std::vector<float> th(nq_pad, t);
      CK(hipMemcpy(dthr, th.data(), nq_pad * 4, hipMemcpyHostToDevice));
      CK(hipMemset(cc, 0, nq * 4));
      batch_score_bf16x3_kernel<METRIC_COS, false, TM, TN, PM><<<b.q_tiles * b.n_tiles, THREADS>>>(b);
      CK(hipMemset(cc, 0, nq * 4));
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < iters; ++i) batch_score_bf16x3_kernel<METRIC_COS, false, TM, TN, PM><<<b.q_tiles * b.n_tiles, THREADS>>>(b);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= iters;
      std::vector<uint32_t> hc(nq);
      CK(hipMemcpy(hc.data(), cc, nq * 4, hipMemcpyDeviceToHost));
      double avg = 0; for (auto v : hc) avg += v; avg /= nq * (double)iters;
      printf("bf16x3 tile %dx%d normal data, thr=%g: %.3f ms  %.1f TFLOP/s f32-equivalent (%.2fx the f32 MFMA peak; %.1f%% of the 2500/3 "
             "bf16 ceiling)  %.0f queries/s  survivors/query %.0f\n", TM, TN, t, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3,
             flop / ms / 1e9 / (2500.0 / 3) * 100, nq / (ms * 1e-3), avg);
    }
  };
  using I64 = std::integral_constant<int, 64>;
  using I128 = std::integral_constant<int, 128>;
  using I256 = std::integral_constant<int, 256>;
  time_tile(I128{}, I128{}, I64{});
  if (nq_pad % 256 == 0) time_tile(I256{}, I256{}, I128{});

  {  // bottleneck probes: 1 = no MFMA (loads + LDS stores + barriers), 2 = no global loads / LDS stores
    std::vector<float> th(nq_pad, -1e30f);
    CK(hipMemcpy(dthr, th.data(), nq_pad * 4, hipMemcpyHostToDevice));
    BatchArgs bb = a;
    bb.q_tiles = nq_pad / 256;
    bb.n_tiles = (n + 255) / 256;
    auto run = [&](int dbg) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, 0));
        const int g = (nq_pad / 256) * ((n + 255) / 256);
        if (dbg == 1) batch_score_bf16x3_kernel<METRIC_COS, false, 256, 256, 128, 1><<<g, 512>>>(bb);
        else if (dbg == 2) batch_score_bf16x3_kernel<METRIC_COS, false, 256, 256, 128, 2><<<g, 512>>>(bb);
        else if (dbg == 3) batch_score_bf16x3_kernel<METRIC_COS, false, 256, 256, 128, 3><<<g, 512>>>(bb);
        else if (dbg == 4) batch_score_bf16x3_kernel<METRIC_COS, false, 256, 256, 128, 4><<<g, 512>>>(bb);
        else batch_score_bf16x3_kernel<METRIC_COS, false, 256, 256, 128, 5><<<g, 512>>>(bb);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      const char *what[] = {"", "no MFMA", "no global loads / LDS stores", "no loads / stores / barriers",
                            "no loads / stores / barriers, one LDS stage only", "MFMAs only (operands from registers)"};
      printf("probe %d on the 256x256 tile (%s): %.3f ms\n", dbg, what[dbg], ms);
    };
    run(1);
    run(2);
    run(3);
    run(4);
    run(5);
  }
  return fails;
}
