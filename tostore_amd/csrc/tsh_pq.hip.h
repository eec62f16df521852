// tsh_pq.hip.h -- write-path helper: batch PQ encode of
// device-resident rows.  Reference arithmetic: batchPqEncode,
// /root/reference/lib/src/core/compute_tasks.dart:2292-2326 (== VectorQuantizer.
// encode / _nearestCentroid, core/vector_quantizer.dart:357-368,461-483):
// per vector and sub-space, the FIRST centroid with the smallest squared
// distance, the distance accumulated in f64 over the sub-space's dimensions in
// order, one rounding per multiply and per add.  The kernel does exactly that in
// f64 (no two-stage trick needed: 8 terms per distance), so codes are bit-exact.
//
// One thread per vector; the codebook (pre-widened to f64 on the host) is read
// with wave-uniform addresses, i.e. through the scalar cache; a thread walks its
// own row left to right across the sub-spaces.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tsh {

struct PqEncodeArgs {
  const float *rows;       // device rows, stride ld
  const double *codebook;  // subspaces x centroids x sub_dim, f64 (exact widening of the f32 codebook)
  uint8_t *codes;          // n x subspaces
  int64_t ld;
  int64_t first;           // first local row
  int64_t n;
  int32_t subspaces, centroids, sub_dim;
};

template <int SD>  // sub-space width known at compile time (0 = runtime, up to 64)
__global__ void __launch_bounds__(256) pq_encode_kernel(PqEncodeArgs a) {
#pragma clang fp contract(off)
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= a.n) return;
  const int sd = SD ? SD : a.sub_dim;
  const float *row = a.rows + (a.first + v) * a.ld;
  uint8_t *out = a.codes + v * a.subspaces;
  for (int m = 0; m < a.subspaces; ++m) {
    double x[SD ? SD : 64];
#pragma unroll
    for (int d = 0; d < (SD ? SD : 64); ++d)
      if (d < sd) x[d] = (double)row[m * sd + d];
    const double *cb = a.codebook + (int64_t)m * a.centroids * sd;
    int best_idx = 0;
    double best = __builtin_inf();
    for (int c = 0; c < a.centroids; ++c) {
      double dist = 0;
#pragma unroll
      for (int d = 0; d < (SD ? SD : 64); ++d)
        if (d < sd) {
          double diff = x[d] - cb[(int64_t)c * sd + d];  // wave-uniform address: scalar load
          dist = dist + diff * diff;
        }
      if (dist < best) {  // strict: ties and NaN keep the earlier centroid
        best = dist;
        best_idx = c;
      }
    }
    out[m] = (uint8_t)best_idx;
  }
}

}  // namespace tsh
