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

// ---- codebook training (N4, second half) -----------------------------------------
// Reference: trainPqSubspace, /root/reference/lib/src/core/compute_tasks.dart:2135-2266,
// dispatched once per sub-space by VectorIndexManager (core/vector_index_manager.dart:
// 740-850) when >= 100 samples were collected.  The only non-arithmetic input is the
// list of k initial sample indices (Dart's Random(42 + subspaceIndex).nextInt(n)), which
// the caller supplies.  Everything else is restated with the widths of the Dart text:
//   norms[c]   = f32( 0.5 * sum_d f64(c_d)^2 )                       (:2166-2174)
//   score(i,c) = dot - f64(norms[c]); first strict maximum           (:2183-2233)
//       dot: subDim % 4 == 0 -> per 4 dims, f32 products summed ((x+y)+z)+w in f64;
//            otherwise f64 products, one add per dim
//   sums       = Float32List, += per member sample in sample order   (:2236-2248)
//   new        = f64(sums) * (1.0 / count) -> f32; moved = |old-new| > 1e-4 (:2250-2261)
//   stop after the update of the first iteration in which nothing moved (:2262)
// All sub-spaces train in one grid (blockIdx.y = sub-space); `active[m]` is cleared
// on the device when a sub-space converges so that no host round trip is needed.
struct PqTrainArgs {
  const float *samples;     // n x dim
  float *centroids;         // subspaces x k x sub_dim
  float *norms;             // subspaces x k
  int32_t *assign;          // subspaces x n
  const int32_t *init_idx;  // subspaces x k
  int32_t *active;          // subspaces
  int32_t *changed;         // subspaces
  int32_t n, dim, subspaces, k, sub_dim;
};

static __global__ void pq_train_init_kernel(PqTrainArgs a) {
  const int m = blockIdx.y, c = blockIdx.x, d = threadIdx.x;
  if (d < a.sub_dim)
    a.centroids[((int64_t)m * a.k + c) * a.sub_dim + d] =
        a.samples[(int64_t)a.init_idx[m * a.k + c] * a.dim + m * a.sub_dim + d];
  if (c == 0 && d == 0) {
    a.active[m] = 1;
    a.changed[m] = 0;
  }
}

static __global__ void __launch_bounds__(256) pq_train_norms_kernel(PqTrainArgs a) {
#pragma clang fp contract(off)
  const int m = blockIdx.x, c = threadIdx.x;
  if (!a.active[m]) return;
  if (c == 0) a.changed[m] = 0;
  if (c >= a.k) return;
  const float *cv = a.centroids + ((int64_t)m * a.k + c) * a.sub_dim;
  double norm = 0;
  for (int d = 0; d < a.sub_dim; ++d) {
    double v = (double)cv[d];
    norm = norm + v * v;
  }
  a.norms[m * a.k + c] = (float)(norm * 0.5);
}

template <int SD, bool SIMD>  // SD = compile-time sub-space width (0: runtime, <= 64)
__global__ void __launch_bounds__(256) pq_train_assign_kernel(PqTrainArgs a) {
#pragma clang fp contract(off)
  const int m = blockIdx.y;
  if (!a.active[m]) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const int sd = SD ? SD : a.sub_dim;
  float x[SD ? SD : 64];
  const float *row = a.samples + (int64_t)i * a.dim + m * sd;
#pragma unroll
  for (int d = 0; d < (SD ? SD : 64); ++d)
    if (d < sd) x[d] = row[d];
  const float *cb = a.centroids + (int64_t)m * a.k * sd;  // wave-uniform addresses: scalar loads
  const float *nr = a.norms + m * a.k;
  double best = -__builtin_inf();
  int best_idx = 0;
  for (int c = 0; c < a.k; ++c) {
    const float *cv = cb + (int64_t)c * sd;
    double dot = 0;
    if (SIMD) {
#pragma unroll
      for (int d = 0; d < (SD ? SD : 64); d += 4)
        if (d < sd) {
          float rx = __fmul_rn(x[d], cv[d]), ry = __fmul_rn(x[d + 1], cv[d + 1]);
          float rz = __fmul_rn(x[d + 2], cv[d + 2]), rw = __fmul_rn(x[d + 3], cv[d + 3]);
          dot = dot + ((((double)rx + (double)ry) + (double)rz) + (double)rw);
        }
    } else {
#pragma unroll
      for (int d = 0; d < (SD ? SD : 64); ++d)
        if (d < sd) dot = dot + (double)x[d] * (double)cv[d];
    }
    const double score = dot - (double)nr[c];
    if (score > best) {  // strict: the first maximum wins, NaN never does
      best = score;
      best_idx = c;
    }
  }
  a.assign[(int64_t)m * a.n + i] = best_idx;
}

// One wave per (centroid, sub-space): lanes own dimensions, members are visited in
// sample order (ballot over 64 samples at a time, then a bit scan), so each lane's
// f32 running sum sees exactly the reference's sequence of additions.
static __global__ void __launch_bounds__(64) pq_train_update_kernel(PqTrainArgs a) {
#pragma clang fp contract(off)
  const int m = blockIdx.y, c = blockIdx.x, lane = threadIdx.x;
  if (!a.active[m]) return;
  const int sd = a.sub_dim;
  const int32_t *as = a.assign + (int64_t)m * a.n;
  const float *col = a.samples + m * sd + (lane < sd ? lane : 0);
  float sum = 0.f;
  int count = 0;
  for (int i0 = 0; i0 < a.n; i0 += 64) {
    const int i = i0 + lane;
    unsigned long long members = __ballot(i < a.n && as[i] == c);
    count += __popcll(members);
    while (members) {
      const int j = __builtin_ctzll(members);
      members &= members - 1;
      sum = (float)((double)sum + (double)col[(int64_t)(i0 + j) * a.dim]);
    }
  }
  if (count == 0 || lane >= sd) return;
  float *cp = a.centroids + ((int64_t)m * a.k + c) * sd + lane;
  const double inv = 1.0 / (double)count;
  const double nv = (double)sum * inv;
  if (fabs((double)*cp - nv) > 1e-4) atomicOr(a.changed + m, 1);
  *cp = (float)nv;
}

static __global__ void pq_train_flag_kernel(PqTrainArgs a) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m < a.subspaces && a.active[m] && !a.changed[m]) a.active[m] = 0;
}

}  // namespace tsh
