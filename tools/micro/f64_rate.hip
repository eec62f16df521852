// Issue rate of the f64 instructions the exact re-rank is made of (gfx950), one wave per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/f64_rate tools/micro/f64_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int OP>
__global__ void __launch_bounds__(64) k(double *out, const float *in, int iters, int active) {
  const int lane = threadIdx.x;
  float f0 = in[lane], f1 = in[lane + 64], f2 = in[lane + 128], f3 = in[lane + 192];
  double a0 = f0, a1 = f1, a2 = f2, a3 = f3, b0 = 1.0000001, b1 = 0.9999999;
  if (lane >= active) return;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (OP == 0) {  // 4 independent cvt f32 -> f64
        asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7"
                     : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));
      } else if (OP == 1) {  // 4 independent fma
        asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
      } else if (OP == 2) {  // 4 independent add
        asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
      } else if (OP == 3) {  // 4 independent mul
        asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
      } else if (OP == 4) {  // 4 DEPENDENT adds (chain latency)
        asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1" : "+v"(a0) : "v"(b0));
      } else if (OP == 5) {  // 4 DEPENDENT fma
        asm volatile("v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %0, %1, %2, %0" : "+v"(a0) : "v"(b0), "v"(b1));
      } else if (OP == 6) {  // 4 independent f32 fma for scale
        asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                     : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(f0), "v"(f1));
      } else if (OP == 7) {  // the re-rank's cosine element: cvt, 2 fma (two chains), x4 elements
        asm volatile("v_cvt_f64_f32 %2, %3\n v_fma_f64 %0, %2, %4, %0\n v_fma_f64 %1, %2, %2, %1" : "+v"(a0), "+v"(a1), "=&v"(a2) : "v"(f0), "v"(b0));
        asm volatile("v_cvt_f64_f32 %2, %3\n v_fma_f64 %0, %2, %4, %0\n v_fma_f64 %1, %2, %2, %1" : "+v"(a0), "+v"(a1), "=&v"(a3) : "v"(f1), "v"(b0));
        asm volatile("v_cvt_f64_f32 %2, %3\n v_fma_f64 %0, %2, %4, %0\n v_fma_f64 %1, %2, %2, %1" : "+v"(a0), "+v"(a1), "=&v"(a2) : "v"(f2), "v"(b0));
        asm volatile("v_cvt_f64_f32 %2, %3\n v_fma_f64 %0, %2, %4, %0\n v_fma_f64 %1, %2, %2, %1" : "+v"(a0), "+v"(a1), "=&v"(a3) : "v"(f3), "v"(b0));
      }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + f0 + f1 + f2 + f3;
  if (lane == 0 && blockIdx.x == 0) reinterpret_cast<uint64_t *>(out)[4096] = t1 - t0;
}

template <int OP>
void run(const char *what, int per_iter, int waves_per_simd, int active) {
  double *out;
  float *in;
  hipMalloc(&out, 8 * 8192);
  hipMalloc(&in, 4 * 256);
  hipMemset(in, 0, 4 * 256);
  const int iters = 2000;
  const int grid = 256 * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<OP><<<grid, 64>>>(out, in, 10, active);
  hipEventRecord(e0);
  k<OP><<<grid, 64>>>(out, in, iters, active);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  uint64_t cyc;
  hipMemcpy(&cyc, reinterpret_cast<uint64_t *>(out) + 4096, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 16 * per_iter;
  printf("%-44s waves/SIMD %d lanes %2d: %7.2f ns per instr per wave (wall), %6.2f counter ticks/instr\n", what, waves_per_simd,
         active, ms * 1e6 / n / waves_per_simd, (double)cyc / n);
  hipFree(out);
  hipFree(in);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0>("v_cvt_f64_f32 (independent)", 4, w, 64);
    run<1>("v_fma_f64 (independent)", 4, w, 64);
    run<2>("v_add_f64 (independent)", 4, w, 64);
    run<3>("v_mul_f64 (independent)", 4, w, 64);
    run<4>("v_add_f64 (dependent chain)", 4, w, 64);
    run<5>("v_fma_f64 (dependent chain)", 4, w, 64);
    run<6>("v_fma_f32 (independent)", 4, w, 64);
    run<7>("cosine element: cvt + 2 fma (2 chains)", 12, w, 64);
  }
  run<1>("v_fma_f64 (independent)", 4, 1, 16);
  run<0>("v_cvt_f64_f32 (independent)", 4, 1, 16);
  run<7>("cosine element: cvt + 2 fma (2 chains)", 12, 1, 16);
  return 0;
}
