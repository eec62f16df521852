// scan_bench.hip -- tuning harness for the K1 scan kernel (not part of the
// product library).  Times template variants of tsh::scan_kernel on synthetic
// data with HIP events and prints achieved algorithmic GB/s (n*d*4 bytes per
// launch), next to a plain read-only streaming kernel as the practical ceiling.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o scan_bench scan_bench.hip
//   ./scan_bench [n=1000000] [d=768] [iters=20]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../tostore_amd/csrc/tsh_kernels.hip.h"

using namespace tsh;

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

__global__ void fill_kernel(float *p, size_t n, uint32_t seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = ((float)(x & 0xFFFFFF) / 8388608.0f - 1.0f) * 0.05f;
  }
}

template <bool NT>
__global__ void __launch_bounds__(256) stream_read_kernel(const float *p, size_t n4, float *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t st = (size_t)gridDim.x * blockDim.x;
  f32x4 acc = {0, 0, 0, 0};
  for (; i + 3 * st < n4; i += 4 * st) {
    f32x4 a = ld16<NT>(p + 4 * i), b = ld16<NT>(p + 4 * (i + st)), c = ld16<NT>(p + 4 * (i + 2 * st)),
          d = ld16<NT>(p + 4 * (i + 3 * st));
    acc += a + b + c + d;
  }
  for (; i < n4; i += st) acc += ld16<NT>(p + 4 * i);
  float s = acc.x + acc.y + acc.z + acc.w;
  if (s == 123.456f) out[0] = s;
}

__global__ void checksum_kernel(const uint32_t *keys, size_t n, unsigned long long *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t st = (size_t)gridDim.x * blockDim.x;
  unsigned long long s = 0;
  for (; i < n; i += st) s += (unsigned long long)keys[i] * (i % 1000003 + 1);
  atomicAdd(out, s);
}

struct Variant {
  std::string name;
  void (*launch)(const ScanArgs &, int grid, hipStream_t);
  int waves;
};

template <int NCH, int METRIC, bool FULL, bool MASKED, int R, bool NT, int WAVES, int MINW>
void launch_v(const ScanArgs &a, int grid, hipStream_t s) {
  static ScanArgsQ aq;  // query passed by pointer here (a.query != NULL), q[] unused
  aq.a = a;
  scan_kernel<NCH, METRIC, FULL, MASKED, R, NT, WAVES, MINW><<<grid, WAVES * 64, 0, s>>>(aq);
}

#define V(NCH, R, NT, WAVES, MINW)                                                        \
  vars.push_back({"nch" #NCH " R" #R " nt" #NT " w" #WAVES " minw" #MINW,                 \
                  launch_v<NCH, METRIC_L2, true, false, R, NT, WAVES, MINW>, WAVES})
#define VM(NCH, R, NT, WAVES, MINW)                                                       \
  mvars.push_back({"MASKED nch" #NCH " R" #R " nt" #NT " w" #WAVES " minw" #MINW,         \
                   launch_v<NCH, METRIC_L2, true, true, R, NT, WAVES, MINW>, WAVES})

int main(int argc, char **argv) {
  size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 1000000;
  int d = argc > 2 ? atoi(argv[2]) : 768;
  int iters = argc > 3 ? atoi(argv[3]) : 20;
  int nch = d / 256;
  if (d % 256 || (nch != 3 && nch != 6 && nch != 1 && nch != 2 && nch != 4)) {
    fprintf(stderr, "d must be 256*{1,2,3,4,6}\n");
    return 1;
  }
  size_t cap = (n + 63) / 64 * 64;
  float *rows, *query, *inv, *sink;
  uint32_t *keys, *gmin;
  uint64_t *live, *mask;
  unsigned long long *csum;
  CK(hipMalloc(&rows, cap * d * 4));
  CK(hipMalloc(&query, d * 4));
  CK(hipMalloc(&inv, cap * 4));
  CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&keys, cap * 4));
  CK(hipMalloc(&gmin, cap / 64 * 4));
  CK(hipMalloc(&live, cap / 64 * 8));
  CK(hipMalloc(&mask, cap / 64 * 8));
  CK(hipMalloc(&csum, 8));
  fill_kernel<<<4096, 256>>>(rows, cap * d, 1u);
  fill_kernel<<<4, 256>>>(query, d, 2u);
  CK(hipMemset(live, 0xFF, cap / 64 * 8));
  CK(hipDeviceSynchronize());
  ScanArgs a;
  a.rows = rows; a.query = query; a.query_out = nullptr; a.inv_norm = inv; a.live = live; a.mask = nullptr;
  a.keys = keys; a.gmin = gmin; a.ld = d; a.n = (int64_t)n; a.d4 = d / 4;
  a.n_tiles = (int)((n + 63) / 64);
  double bytes = (double)n * d * 4;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device %s CUs=%d  n=%zu d=%d bytes/launch=%.3f GB iters=%d\n", prop.name, cus, n, d, bytes / 1e9, iters);

  auto time_it = [&](auto fn) {
    fn(); fn();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) fn();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return (double)ms * 1e3 / iters;
  };
  for (int g : {cus * 4, cus * 8, cus * 16, cus * 32}) {
    double us0 = time_it([&] { stream_read_kernel<false><<<g, 256>>>(rows, cap * d / 4, sink); });
    double us1 = time_it([&] { stream_read_kernel<true><<<g, 256>>>(rows, cap * d / 4, sink); });
    printf("stream_read grid=%5d  plain %8.1f us %7.1f GB/s | nt %8.1f us %7.1f GB/s\n", g, us0,
           bytes / us0 / 1e3, us1, bytes / us1 / 1e3);
  }

  std::vector<Variant> vars, mvars;
  if (nch == 3) {
    V(3, 4, true, 4, 4); V(3, 4, false, 4, 4); V(3, 4, true, 4, 3); V(3, 4, true, 4, 2);
    V(3, 2, true, 4, 4); V(3, 2, true, 4, 5); V(3, 2, false, 4, 5); V(3, 2, true, 4, 3);
    V(3, 4, true, 1, 4); V(3, 4, true, 2, 4); V(3, 4, true, 8, 4); V(3, 2, true, 1, 5);
    V(3, 2, true, 2, 5); V(3, 2, true, 8, 5); V(3, 4, true, 8, 2); V(3, 4, true, 1, 2);
    VM(3, 2, true, 4, 4); VM(3, 2, true, 4, 5); VM(3, 4, true, 4, 3);
  } else if (nch == 6) {
    V(6, 2, true, 4, 3); V(6, 2, false, 4, 3); V(6, 2, true, 4, 2); V(6, 2, true, 8, 2); V(6, 2, true, 2, 3);
    V(6, 4, true, 4, 2); V(6, 4, true, 4, 1);
  } else if (nch == 1) {
    V(1, 4, true, 4, 4); V(1, 4, true, 4, 8); V(1, 2, true, 4, 8); V(1, 4, false, 4, 8);
  } else if (nch == 2) {
    V(2, 4, true, 4, 4); V(2, 4, true, 4, 5); V(2, 2, true, 4, 8); V(2, 2, true, 4, 5);
  } else {
    V(4, 2, true, 4, 4); V(4, 4, true, 4, 2); V(4, 4, true, 4, 3); V(4, 2, false, 4, 4);
  }
  unsigned long long ref = 0;
  for (auto &v : vars) {
    for (int mode = 0; mode < 4; ++mode) {
      int full = (a.n_tiles + v.waves - 1) / v.waves;
      int grid = mode == 0 ? full : cus * (mode == 1 ? 4 : mode == 2 ? 8 : 16) * 4 / v.waves;
      if (mode > 0 && grid >= full) continue;
      CK(hipMemset(keys, 0, cap * 4));
      double us = time_it([&] { v.launch(a, grid, 0); });
      CK(hipMemset(csum, 0, 8));
      checksum_kernel<<<1024, 256>>>(keys, n, csum);
      unsigned long long cs;
      CK(hipMemcpy(&cs, csum, 8, hipMemcpyDeviceToHost));
      if (!ref) ref = cs;
      printf("%-28s grid=%6d %8.1f us %7.1f GB/s  %5.1f%% of 8TB/s %s\n", v.name.c_str(), grid, us,
             bytes / us / 1e3, bytes / us / 1e3 / 80.0, cs == ref ? "" : "CHECKSUM MISMATCH");
    }
  }
  // masked variants: all-live mask (same answer), then ~50% and ~10% Bernoulli masks
  std::vector<uint64_t> hm(cap / 64);
  for (int pct : {100, 50, 10}) {
    uint32_t x = 12345u;
    for (auto &w : hm) {
      w = 0;
      for (int b = 0; b < 64; ++b) {
        x = x * 1664525u + 1013904223u;
        if ((x >> 8) % 100 < (uint32_t)pct) w |= 1ull << b;
      }
    }
    CK(hipMemcpy(mask, hm.data(), hm.size() * 8, hipMemcpyHostToDevice));
    a.mask = mask;
    for (auto &v : mvars) {
      int grid = (a.n_tiles + v.waves - 1) / v.waves;
      double us = time_it([&] { v.launch(a, grid, 0); });
      CK(hipMemset(csum, 0, 8));
      checksum_kernel<<<1024, 256>>>(keys, n, csum);
      unsigned long long cs;
      CK(hipMemcpy(&cs, csum, 8, hipMemcpyDeviceToHost));
      printf("%-34s keep=%3d%% %8.1f us  useful %7.1f GB/s %s\n", v.name.c_str(), pct, us,
             bytes * pct / 100 / us / 1e3, pct == 100 ? (cs == ref ? "ok" : "CHECKSUM MISMATCH") : "");
    }
  }
  return 0;
}
