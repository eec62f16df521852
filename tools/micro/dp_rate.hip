// Round 5 micro-benchmark (beside f64_rate.hip of round 3): what f64 VALU instructions cost a gfx950 SIMD with ONE wave
// on it -- latency of a dependent chain and issue rate of four independent ones (v_add_f64, v_mul_f64, v_fma_f64,
// v_cvt_f64_f32).  Read on one MI355X: dependent add / fma 4.1-4.3 ns, independent 2.3-2.4 ns, cvt 3.8 ns per instruction.
// hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o dp_rate dp_rate.hip && ./dp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP, int CHAINS>
__global__ void __launch_bounds__(256) k(double *out, uint64_t *cyc, int iters, double seed) {
#pragma clang fp contract(off)
  double x[CHAINS];
  float f = (float)seed + threadIdx.x;
  for (int c = 0; c < CHAINS; ++c) x[c] = seed + c + threadIdx.x;
  const double y = seed * 0.5 + 1.0;
  __syncthreads();
  const uint64_t t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        if (OP == 0) x[c] = x[c] + y;
        else if (OP == 1) x[c] = x[c] * y;
        else if (OP == 2) x[c] = __builtin_fma(x[c], y, y);
        else {
          asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x[c]) : "v"(f));
        }
      }
    }
  }
  const uint64_t t1 = clock64();
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP, int CHAINS>
void run(const char *name, int waves_per_simd) {
  double *out;
  uint64_t *cyc;
  const int blocks = 256;  // one workgroup per CU (as far as the dispatcher spreads them), 4 * waves_per_simd waves each
  const int threads = 256 * waves_per_simd;
  hipMalloc(&out, sizeof(double) * blocks * 1024);
  hipMalloc(&cyc, sizeof(uint64_t) * blocks);
  const int iters = 256;
  hipLaunchKernelGGL((k<OP, CHAINS>), dim3(blocks), dim3(threads > 1024 ? 1024 : threads), 0, 0, out, cyc, iters, 1.0);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP, CHAINS>), dim3(blocks), dim3(threads > 1024 ? 1024 : threads), 0, 0, out, cyc, iters, 1.0);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  uint64_t h[256];
  hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  double avg = 0;
  for (int i = 0; i < blocks; ++i) avg += (double)h[i];
  avg /= blocks;
  const double instr = (double)iters * 16 * CHAINS;  // per wave
  printf("%-14s chains %d, %d wave(s)/SIMD: %7.2f counter ticks per instruction per wave; kernel %.1f us -> %.2f ns per instruction per wave\n",
         name, CHAINS, waves_per_simd, avg / instr, ms * 1e3, ms * 1e6 / instr);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 1; ++w) {  // (more waves per SIMD need more than one workgroup per CU: not what this measures)
    run<0, 1>("v_add_f64", w);
    run<0, 4>("v_add_f64", w);
    run<1, 4>("v_mul_f64", w);
    run<2, 1>("v_fma_f64", w);
    run<2, 4>("v_fma_f64", w);
    run<3, 4>("v_cvt_f64_f32", w);
  }
  return 0;
}
