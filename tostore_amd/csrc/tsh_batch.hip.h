// tsh_batch.hip.h -- batched-query path: Q (nq x d) against the resident rows
// as a dense contraction on the FP32 matrix cores (v_mfma_f32_32x32x2_f32,
// exact f32, 157 TF/s peak on MI355X), with the ranking-key transform and a
// per-query threshold filter fused into the epilogue so the nq x n score
// matrix never reaches HBM.
//
// Exactness comes from the same two-stage scheme as the single-query path:
//   B0  score a SAMPLE of the rows densely (rows [0, n_sample)); per query the
//       k-th smallest sample key is a proven upper bound tau_q on the k-th
//       smallest key over ALL rows (the sample alone holds k rows <= tau_q)
//   B1  score everything else with the filter key <= band(tau_q): about
//       k * n / n_sample survivors per query, appended to per-query lists
//   B2  per query: exact k-th smallest key of its list -> band -> candidates
//   K4  the f64 rerank of tsh_kernels.hip.h, one (query, candidate) per wave
// Keys: IP -dot; cosine -dot/|v|; L2 |q|^2 + |v|^2 - 2 dot.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "tsh_kernels.hip.h"

namespace tsh {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int BT_M = 128;   // queries per workgroup tile
constexpr int BT_N = 128;   // rows per workgroup tile
constexpr int BT_K = 32;    // default reduction chunk (template parameter BK of the kernel)
// LDS row stride in floats for a chunk of BK: BK + 4 (144 B / 80 B rows keep ds_read_b128 conflict-free)
constexpr int bt_ld(int bk) { return bk + 4; }
constexpr int BT_THREADS = 256;

struct BatchArgs {
  const float *Q;         // nq_pad x ld (rows past nq are zero)
  const float *V;         // corpus rows, n x ld
  const float *inv_norm;  // cosine
  const float *sqnorm;    // L2: |v|^2
  const float *qsq;       // L2: |q|^2 per query (nq_pad)
  const float *thr;       // filter mode: per-query band (float, key space); nq_pad
  const float *kmax;      // f16 ping-pong kernel, IP / cosine: the largest key a row can have for this query (nq_pad)
  const float *alpha;     // f16 ping-pong kernel, L2 / IP (nullable): per-row part of the key's error band, the band being
                          // alpha_q |v| + beta_q; the filtered pass tests and stores key - alpha_q |v| (nq_pad)
  const uint64_t *live;   // nullable: bit = row present & not deleted
  const uint64_t *mask;   // nullable: caller keep mask
  float *dense;           // dense mode: nq_pad x dense_ld keys of rows [row0,row1)
  uint32_t *cand_key;     // filter mode: nq x cand_cap (float bits of the key)
  uint32_t *cand_row;     // filter mode: nq x cand_cap
  uint32_t *cand_cnt;     // filter mode: nq (may exceed cand_cap: overflow)
  int64_t ld;
  int64_t dense_ld;
  int32_t row0, row1;     // rows scored by this launch
  int32_t nq, nq_pad;     // nq_pad multiple of BT_M
  int32_t kchunks;        // ceil(ld / BK); ld is padded with zeros to a multiple of 4
  int32_t cand_cap;
  int32_t q_tiles;        // nq_pad / BT_M
  int32_t n_tiles;        // ceil((row1-row0) / BT_N)
  // bf16x3 variant: operands pre-split into bf16 (hi, lo) planes, see split_rows_kernel
  const u32x4 *Qs;        // nq_pad x hchunks x 128 B, see plane_piece()
  const u32x4 *Vs;        // round_up(n, 256) x hchunks x 128 B
  int32_t hchunks;        // ceil(dim / 32)
  int32_t tile_m;         // workgroup tile (queries = rows): 128 or 256 (host-side dispatch only)
  float dot_scale;        // f16 variant: 2^-(eq + ev), undoes the power-of-two operand scales (0 = unused)
  int32_t dbg;            // f16 kernel probe (TSH_F16_DBG): 4 = no epilogue (results are wrong), 32 = timestamps
  uint64_t *dbg_buf;      // TSH_F16_DBG & 32: [wave 0 / wave 4 of workgroup 0][step][point] shader-clock stamps
  const uint32_t *row_ids;  // f16 ping-pong kernel, nullable: the plane holds its rows in ANOTHER ORDER than the row store
                            // (a norm-grouped plane, plane_group_kernel; the hub rows' gathered copy): plane position p
                            // is row row_ids[p] of the shard -- where its live / mask bit is, and the id a survivor is
                            // listed under.  `sqnorm` is then indexed by position too.
};

// an upper bound of |v| from the stored |v|^2 (f32 from an f64 sum: 2^-24; the square root: one ulp)
__device__ __forceinline__ float batch_norm_up(float sq) { return __builtin_sqrtf(sq) * 1.000001f; }

// XCD-aware tile order: workgroup b runs on XCD b % 8; give each XCD runs of
// q_tiles consecutive workgroups that share one row tile (L2 reuse of V)
// Per-query candidate counters sit one per 128-byte line: every survivor of the filtered pass is one returning
// atomic on its query's counter, and packed counters (a small batch: all of them in a few lines, i.e. a few L2
// channels) serialise there.
constexpr int CC_STRIDE = 32;  // uint32 units (4 KiB apart measured the same)

__device__ __forceinline__ void batch_tile_of(const BatchArgs &a, int b, int *q_tile, int *n_tile) {
  const int total = a.q_tiles * a.n_tiles;
  const int xcd = b & 7, i = b >> 3;
  const int per_round = 8 * a.q_tiles;
  const int full_rounds = total / per_round;
  if (b < full_rounds * per_round) {
    *q_tile = i % a.q_tiles;
    *n_tile = (i / a.q_tiles) * 8 + xcd;
  } else {  // ragged tail: plain order
    int r = b - full_rounds * per_round;
    *n_tile = full_rounds * 8 + r / a.q_tiles;
    *q_tile = r % a.q_tiles;
  }
}

// Epilogue shared by the f32 and the bf16x3 kernels: key transform (+ filter) of a wave's
// 64 x 64 patch held as 2 x 2 MFMA 32x32 accumulators.
template <int METRIC, bool DENSE, bool SCALED = false>
__device__ __forceinline__ void batch_epilogue(const BatchArgs &a, f32x16 (&acc)[2][2], const float *s_thr,
                                               const float *s_qsq, int qbase, int nbase, int prow, int pcol, int lane) {
  // ---- epilogue: key transform (+ filter) -----------------------------------------
  // C/D map of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5):
  // for a fixed reg >> 2 the four rows are consecutive, so a lane's 16 per-row
  // values (threshold, |q|^2) are four float4 reads.
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rbase = prow + i * 32 + 4 * (lane >> 5);  // tile row of reg 0
    f32x4 th[4], qq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      th[g] = *reinterpret_cast<const f32x4 *>(&s_thr[rbase + 8 * g]);
      if (METRIC == METRIC_L2) qq[g] = *reinterpret_cast<const f32x4 *>(&s_qsq[rbase + 8 * g]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = nbase + pcol + j * 32 + (lane & 31);  // corpus row
      const bool col_ok = col < a.row1;
      float vin = 0.f, vsq = 0.f;
      bool alive = col_ok;
      if (col_ok) {
        if (METRIC == METRIC_COS) vin = a.inv_norm ? a.inv_norm[col] : 1.f;  // f16 planes hold unit rows
        if (METRIC == METRIC_L2) vsq = a.sqnorm[col];
        if (a.live) alive = (a.live[col >> 6] >> (col & 63)) & 1ull;
        if (alive && a.mask) alive = (a.mask[col >> 6] >> (col & 63)) & 1ull;
      }
      float key[16];
      uint32_t pass = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float dot = SCALED ? acc[i][j][r] * a.dot_scale : acc[i][j][r];  // power of two: exact
        if (METRIC == METRIC_IP) key[r] = -dot;
        else if (METRIC == METRIC_COS) key[r] = -(dot * vin);
        else key[r] = qq[r >> 2][r & 3] + vsq - 2.f * dot;
        if (!DENSE) pass |= (key[r] <= th[r >> 2][r & 3]) ? (1u << r) : 0u;
      }
      if (DENSE) {
        if (col_ok) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int qi = rbase + (r & 3) + 8 * (r >> 2);
            a.dense[(int64_t)(qbase + qi) * a.dense_ld + (col - a.row0)] = alive ? key[r] : __builtin_nanf("");
          }
        }
      } else {
        if (!alive) pass = 0;
        while (pass) {  // rare: about k * n / n_sample survivors per query over the whole pass
          const int r = __builtin_ctz(pass);
          pass &= pass - 1;
          const int q = qbase + rbase + (r & 3) + 8 * (r >> 2);
          if (q < a.nq) {
            float kv = 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) kv = u == r ? key[u] : kv;  // register select, no scratch
            uint32_t p = atomicAdd(&a.cand_cnt[(int64_t)q * CC_STRIDE], 1u);
            if (p < (uint32_t)a.cand_cap) {
              a.cand_key[(int64_t)q * a.cand_cap + p] = __float_as_uint(kv);
              a.cand_row[(int64_t)q * a.cand_cap + p] = (uint32_t)col;
            }
          }
        }
      }
    }
  }
}

// One workgroup = 128 queries x 128 rows; 4 waves as 2 x 2, each wave a
// 64 x 64 patch = 2 x 2 MFMA blocks of 32 x 32.  Both operands are K-major in
// memory; a lane stages float4s through registers into padded LDS, and reads
// its fragments back as one ds_read_b128 per 4 MFMAs: lanes 0-31 own
// k = 8t..8t+3, lanes 32-63 own k = 8t+4..8t+7 (the k order inside a chunk is
// free as long as A and B agree).
template <int METRIC, bool DENSE, int BK = BT_K>
__global__ void __launch_bounds__(BT_THREADS, BK == 32 ? 2 : 3) batch_score_kernel(BatchArgs a) {
  constexpr int BT_LD = bt_ld(BK);
  constexpr int C4 = BK / 4;             // float4 columns per chunk row
  constexpr int RPP = BT_THREADS / C4;   // rows staged per pass
  constexpr int NJ = BT_M / RPP;         // passes per operand
  __shared__ __attribute__((aligned(16))) float As[2][BT_M][BT_LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BT_N][BT_LD];
  __shared__ __attribute__((aligned(16))) float s_thr[BT_M];
  __shared__ __attribute__((aligned(16))) float s_qsq[BT_M];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  int n_tile, q_tile;
  batch_tile_of(a, blockIdx.x, &q_tile, &n_tile);
  const int qbase = q_tile * BT_M;
  const int nbase = a.row0 + n_tile * BT_N;

  if (tid < BT_M) {
    s_thr[tid] = (DENSE || qbase + tid >= a.nq) ? -__builtin_inff() : a.thr[qbase + tid];
    s_qsq[tid] = METRIC == METRIC_L2 ? a.qsq[qbase + tid] : 0.f;
  }

  // staging map: thread -> (row r0 + RPP j, float4 column c4)
  const int c4 = tid % C4, r0 = tid / C4;
  const float *qg[NJ], *vg[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    qg[j] = a.Q + (int64_t)(qbase + r0 + RPP * j) * a.ld + 4 * c4;
    int vr = nbase + r0 + RPP * j;
    if (vr >= a.row1) vr = a.row1 - 1;  // clamp: tail columns are discarded in the epilogue
    vg[j] = a.V + (int64_t)vr * a.ld + 4 * c4;
  }
  const int kmax4 = (int)(a.ld / 4);  // float4s per row

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[NJ], rb[NJ];
  auto gload = [&](int kc) {
    const bool ok = kc * C4 + c4 < kmax4;  // ld need not be a multiple of BK
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      ra[j] = ok ? *reinterpret_cast<const f32x4 *>(qg[j] + kc * BK) : f32x4{0.f, 0.f, 0.f, 0.f};
      rb[j] = ok ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(vg[j] + kc * BK))
                 : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      *reinterpret_cast<f32x4 *>(&As[buf][r0 + RPP * j][4 * c4]) = ra[j];
      *reinterpret_cast<f32x4 *>(&Bs[buf][r0 + RPP * j][4 * c4]) = rb[j];
    }
  };

  gload(0);
  lstore(0);
  __syncthreads();
  const int arow = wm * 64 + (lane & 31), brow = wn * 64 + (lane & 31);
  const int koff = 4 * (lane >> 5);
  for (int kc = 0; kc < a.kchunks; ++kc) {
    const int buf = kc & 1;
    if (kc + 1 < a.kchunks) gload(kc + 1);  // in flight while this chunk is multiplied
#pragma unroll
    for (int t = 0; t < BK / 8; ++t) {
      f32x4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = *reinterpret_cast<const f32x4 *>(&As[buf][arow + 32 * i][8 * t + koff]);
        fb[i] = *reinterpret_cast<const f32x4 *>(&Bs[buf][brow + 32 * i][8 * t + koff]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    }
    if (kc + 1 < a.kchunks) {
      lstore(buf ^ 1);
      __syncthreads();
    }
  }

  batch_epilogue<METRIC, DENSE>(a, acc, s_thr, s_qsq, qbase, nbase, wm * 64, wn * 64, lane);
}

// ---------------------------------------------------------------------------
// bf16x3 variant.  x = hi + lo + r with hi = bf16(x), lo = bf16(x - hi), |r| <= 2^-18 |x|:
//   q.x ~= qh.xh + qh.xl + ql.xh      (three bf16 MFMAs; products exact in f32)
// misses ql.xl + q.r_x + r_q.x <= 3.1 * 2^-18 |q||x| per element, which is an eighth of the
// f32 accumulation bound the band already carries -- the keys are a
// pre-filter, the f64 rerank decides.  v_mfma_f32_32x32x16_bf16 runs 16x the rate of the f32
// MFMA, so three of them per k are 5.3x faster than the f32 kernel at the same tile shape.
//
// Operand layout ("split planes"): per row, per chunk of 32 k-values, 128 bytes:
//   [ hi(k0..k0+31) : 64 B ][ lo(k0..k0+31) : 64 B ],  rows zero-padded to whole chunks
// i.e. 4 B per element like the f32 rows, one full cache line per (row, chunk).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Plane layout, in 16-byte pieces: rows are grouped by 256 and a group's chunk is contiguous,
//   piece(row, chunk, p) = (((row / 256) * hchunks + chunk) * 256 + row % 256) * 8 + p,
// so the 128 / 256 rows a workgroup stages for one chunk are one 16 / 32 KB run.  (With the rows of a
// tile 1.5-3 KB apart -- row-major planes -- every line of a chunk fell on 2-4 of the 16 L2 channels.)
constexpr int PLANE_GROUP = 256;
__host__ __device__ __forceinline__ int64_t plane_piece(int64_t row, int chunk, int hchunks) {
  return (((row / PLANE_GROUP) * hchunks + chunk) * PLANE_GROUP + row % PLANE_GROUP) * 8;
}

__device__ __forceinline__ uint32_t bf16_rne_bits(float f) {  // finite inputs (|x| <= 1e15 outside safe mode)
  uint32_t u = __float_as_uint(f);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

struct SplitArgs {
  const float *rows;  // n x ld f32
  u32x4 *out;         // n x hchunks x 8 pieces of 16 B
  int64_t ld;
  int64_t first, n;   // rows [first, first + n)
  int32_t dim, hchunks;
};

// one thread = 8 consecutive k of one row: a 16-B hi piece and a 16-B lo piece
static __global__ void __launch_bounds__(256) split_rows_kernel(SplitArgs a) {
  const int64_t per_row = (int64_t)a.hchunks * 4;
  const int64_t total = a.n * per_row, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t row = a.first + i / per_row;
    const int p = (int)(i % per_row), kc = p >> 2, c = p & 3;
    const int k0 = kc * 32 + c * 8;
    const float *src = a.rows + row * a.ld + k0;
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      uint32_t h2[2], l2[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int k = k0 + 2 * e + u;
        const float x = k < a.dim ? src[2 * e + u] : 0.f;
        const uint32_t hb = bf16_rne_bits(x);
        const float rest = x - __uint_as_float(hb << 16);  // exact in f32
        h2[u] = hb;
        l2[u] = bf16_rne_bits(rest);
      }
      hi[e] = h2[0] | (h2[1] << 16);
      lo[e] = l2[0] | (l2[1] << 16);
    }
    u32x4 *dst = a.out + plane_piece(row, kc, a.hchunks);
    dst[c] = u32x4{hi[0], hi[1], hi[2], hi[3]};
    dst[4 + c] = u32x4{lo[0], lo[1], lo[2], lo[3]};
  }
}

// f16 variant (TSH_OPT_BATCH_KERNEL = 2): ONE v_mfma_f32_32x32x16_f16 per product on operands rounded
// to fp16 (11-bit significand: 2^-11 per operand, products exact in f32).  Operand layout: per row, per
// chunk of 64 k-values, 128 bytes of fp16 -- 2 B per element.  Values are scaled by a power of two
// (exact) so the largest magnitude sits at 2^13..2^14, far from fp16's 65504 and with 27 binades above
// its subnormal step; cosine rows are stored normalised (x * inv_norm), so their key is -dot itself.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct HalfArgs {
  const float *rows;      // n x ld f32
  const float *inv_norm;  // nullable: multiply each row by its 1/|row| first (cosine corpus)
  u32x4 *out;             // n x hchunks x 8 pieces of 16 B
  int64_t ld;
  int64_t first, n;
  int32_t dim, hchunks;   // hchunks = ceil(dim / 64)
  float scale;            // power of two
};

// one thread = 8 consecutive k of one row = one 16-B piece
static __global__ void __launch_bounds__(256) half_rows_kernel(HalfArgs a) {
  const int64_t per_row = (int64_t)a.hchunks * 8;
  const int64_t total = a.n * per_row, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t row = a.first + i / per_row;
    const int p = (int)(i % per_row);
    const int k0 = p * 8;
    const float *src = a.rows + row * a.ld + k0;
    const float mul = a.inv_norm ? a.inv_norm[row] * a.scale : a.scale;
    f16x8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = (_Float16)(k0 + e < a.dim ? src[e] * mul : 0.f);  // round to nearest even
    a.out[plane_piece(row, p >> 3, a.hchunks) + (p & 7)] = __builtin_bit_cast(u32x4, h);
  }
}

// Tile shapes (TM queries x TN rows per workgroup, each wave a PM x 64 patch of 32 x 32 MFMA blocks):
//   small  128 x 128, PM = 64, 4 waves, two workgroups per CU  -- batches of up to 128 queries
//   big    256 x 256, PM = 128, 8 waves, one workgroup per CU  -- everything larger
// Measured on the small tile (1 M x 768, 1024 queries): operand traffic alone (no MFMA) 3.3 ms, MFMA +
// LDS reads alone 3.0 ms, together 4.65 ms.  The big tile moves half the bytes per flop (L2 -> LDS) and
// reads 25 % fewer LDS bytes per MFMA.
// LDS holds [stage][row][8 pieces of 16 B] (hi pieces 0-3, lo pieces 4-7 of a row's 128-byte line) without
// padding; piece p of row r sits at p ^ ((r >> 1) & 7): the eight lanes that store one row fill its 128
// bytes (all 32 banks once), and the 16-lane groups of ds_read_b128 (MI355X_MICROARCH.md, LDS table) hit
// 16 distinct 16-B bank quads.  A lane's fragment for the k16 slab s is piece 2s + (lane >> 5) of the hi
// or lo half.  Global loads run NSETS - 1 ... NSETS chunks ahead of the multiply in NSETS register sets.
template <int TM, int TN, int PM, int MODE = 0>
struct HbTile {
  static constexpr int WM = TM / PM, WN = TN / 64, WAVES = WM * WN, THREADS = 64 * WAVES;
  static constexpr int NA = TM * 8 / THREADS, NB = TN * 8 / THREADS;  // 16-B pieces a thread stages per chunk
  static constexpr int MI = PM / 32;                                  // MFMA block rows per wave
  // the f16 kernel keeps fewer fragments, so the big tile affords a second register set as well
  static constexpr int NSETS = (PM == 64 || MODE == 1) ? 2 : 1;
  static constexpr int MIN_WG = (PM == 64 && THREADS <= 256) ? 2 : 1;
};

// MODE 0: bf16x3 (chunk = 32 k: 4 hi + 4 lo pieces, three MFMAs per block and slab of 16 k);
// MODE 1: f16 (chunk = 64 k: 8 pieces, one MFMA per block and slab).
template <int METRIC, bool DENSE, int TM = 128, int TN = 128, int PM = 64, int DBG = 0, int MODE = 0>
__global__ void __launch_bounds__((HbTile<TM, TN, PM>::THREADS), (HbTile<TM, TN, PM>::MIN_WG))
    batch_score_bf16x3_kernel(BatchArgs a) {
  using T = HbTile<TM, TN, PM, MODE>;
  constexpr int THREADS = T::THREADS, NA = T::NA, NB = T::NB, MI = T::MI, NSETS = T::NSETS;
  __shared__ u32x4 As[2][TM][8];
  __shared__ u32x4 Bs[2][TN][8];
  __shared__ __attribute__((aligned(16))) float s_thr[TM];
  __shared__ __attribute__((aligned(16))) float s_qsq[TM];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / T::WN, wn = wave % T::WN;
  int n_tile, q_tile;
  batch_tile_of(a, blockIdx.x, &q_tile, &n_tile);
  const int qbase = q_tile * TM;
  const int nbase = a.row0 + n_tile * TN;
  for (int t = tid; t < TM; t += THREADS) {
    s_thr[t] = (DENSE || qbase + t >= a.nq) ? -__builtin_inff() : a.thr[qbase + t];
    s_qsq[t] = METRIC == METRIC_L2 ? a.qsq[qbase + t] : 0.f;
  }

  // staging: piece p = tid + THREADS j of an operand tile's rows x 8 pieces (4 hi, 4 lo);
  // 8 consecutive threads fetch one row's 128-byte line
  const u32x4 *qg[NA], *vg[NB];
  int sa_row[NA], sa_col[NA], sb_row[NB], sb_col[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int p = tid + THREADS * j, r = p >> 3, c8 = p & 7;
    sa_row[j] = r;
    sa_col[j] = c8 ^ ((r >> 1) & 7);
    qg[j] = a.Qs + plane_piece(qbase + r, 0, a.hchunks) + c8;
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int p = tid + THREADS * j, r = p >> 3, c8 = p & 7;
    sb_row[j] = r;
    sb_col[j] = c8 ^ ((r >> 1) & 7);
    int vr = nbase + r;
    if (vr >= a.row1) vr = a.row1 - 1;  // clamp: tail columns are discarded in the epilogue
    vg[j] = a.Vs + plane_piece(vr, 0, a.hchunks) + c8;
  }

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4 ra[NSETS][NA], rb[NSETS][NB];
  auto gload = [&](auto SET, int kc) {
    constexpr int S = decltype(SET)::value;
#pragma unroll
    for (int j = 0; j < NA; ++j) ra[S][j] = qg[j][kc * (PLANE_GROUP * 8)];
#pragma unroll
    for (int j = 0; j < NB; ++j) rb[S][j] = vg[j][kc * (PLANE_GROUP * 8)];  // through L2: other q-tiles re-read it
  };
  auto lstore = [&](auto SET, int buf) {
    constexpr int S = decltype(SET)::value;
#pragma unroll
    for (int j = 0; j < NA; ++j) As[buf][sa_row[j]][sa_col[j]] = ra[S][j];
#pragma unroll
    for (int j = 0; j < NB; ++j) Bs[buf][sb_row[j]][sb_col[j]] = rb[S][j];
  };
  // quarter `part` of a set's LDS stores (the f16 kernel spreads them over the four k16 slabs of the multiply)
  auto lstore_part = [&](auto SET, int buf, int part) {
    constexpr int S = decltype(SET)::value;
#pragma unroll
    for (int j = 0; j < NA; ++j)
      if ((j & 3) == part) As[buf][sa_row[j]][sa_col[j]] = ra[S][j];
#pragma unroll
    for (int j = 0; j < NB; ++j)
      if ((j & 3) == part) Bs[buf][sb_row[j]][sb_col[j]] = rb[S][j];
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, NSETS - 1>;

  const int arow = wm * PM + (lane & 31), brow = wn * 64 + (lane & 31);
  const int half = lane >> 5;
  // (row >> 1) & 7 is the same for row and row + 32
  const int asw = (arow >> 1) & 7, bsw = (brow >> 1) & 7;
  auto multiply = [&](int buf, auto &&mid) {
    if (MODE == 1) {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int pc = 2 * s4 + half;
        f16x8 fa[MI], fb[2];
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[i] = __builtin_bit_cast(f16x8, As[buf][arow + 32 * i][pc ^ asw]);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = __builtin_bit_cast(f16x8, Bs[buf][brow + 32 * j][pc ^ bsw]);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        mid(s4);
      }
      return;
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int pc = 2 * s2 + half;
      bf16x8 ah[MI], al[MI], bh[2], bl[2];
      if (DBG == 5) {  // probe: no LDS reads in the loop, operands = whatever the accumulators hold
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          ah[i] = __builtin_bit_cast(bf16x8, f32x4{acc[i][0][0], acc[i][0][1], acc[i][0][2], acc[i][0][3]});
          al[i] = __builtin_bit_cast(bf16x8, f32x4{acc[i][1][0], acc[i][1][1], acc[i][1][2], acc[i][1][3]});
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bh[j] = __builtin_bit_cast(bf16x8, f32x4{acc[0][j][4], acc[0][j][5], acc[0][j][6], acc[0][j][7]});
          bl[j] = __builtin_bit_cast(bf16x8, f32x4{acc[1][j][4], acc[1][j][5], acc[1][j][6], acc[1][j][7]});
        }
      } else {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        ah[i] = __builtin_bit_cast(bf16x8, As[buf][arow + 32 * i][pc ^ asw]);
        al[i] = __builtin_bit_cast(bf16x8, As[buf][arow + 32 * i][(4 + pc) ^ asw]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bh[j] = __builtin_bit_cast(bf16x8, Bs[buf][brow + 32 * j][pc ^ bsw]);
        bl[j] = __builtin_bit_cast(bf16x8, Bs[buf][brow + 32 * j][(4 + pc) ^ bsw]);
      }
      }
      // product type outermost: MFMAs into the same accumulator are 2 MI issues apart
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    }
  };
  const int nk = a.hchunks;
  if (NSETS == 2) {
    // one step: chunk kc is in LDS stage kc & 1, chunk kc + 1 in flight in the OTHER register set
    auto step = [&](auto CUR, auto NXT, int kc) {
      if (DBG != 2 && kc + 2 < nk) gload(CUR, kc + 2);  // set CUR held chunk kc, already stored to LDS
      if (DBG != 1) multiply(kc & 1, [](int) {});
      if (kc + 1 < nk) {
        if (DBG != 2) lstore(NXT, (kc + 1) & 1);
        __syncthreads();
      }
    };
    // steady state without branches: the compiler's s_waitcnt insertion can then COUNT the loads in flight
    // (vmcnt(8 ...) before the LDS stores of the older set); behind an `if` it assumes the worst at the join
    // and drains everything, which cuts the run-ahead from two chunks to one and parks the waves
    auto fstep = [&](auto CUR, auto NXT, int kc) {
      if (DBG != 2) gload(CUR, kc + 2);
      __builtin_amdgcn_sched_barrier(0);  // keep the loads AHEAD of the multiply (the scheduler sinks them behind it)
      if (MODE == 1 && DBG == 0) {
        // the other set's LDS stores ride between the slabs: the store path (13 cycles per ds_write_b128) works
        // while the matrix pipe still has the slab's MFMAs queued, instead of all waves storing at once
        multiply(kc & 1, [&](int s4) {
          lstore_part(NXT, (kc + 1) & 1, s4);
          __builtin_amdgcn_sched_barrier(0);
        });
      } else {
        if (DBG != 1) multiply(kc & 1, [](int) {});
        if (DBG != 2) lstore(NXT, (kc + 1) & 1);
      }
      __syncthreads();
    };
    gload(S0{}, 0);
    lstore(S0{}, 0);
    gload(S1{}, nk > 1 ? 1 : 0);  // (a one-chunk row re-reads chunk 0: never stored)
    __syncthreads();
    int kc = 0;
    for (; kc + 3 < nk; kc += 2) {
      fstep(S0{}, S1{}, kc);
      fstep(S1{}, S0{}, kc + 1);
    }
    for (; kc < nk; kc += 2) {
      step(S0{}, S1{}, kc);
      if (kc + 1 < nk) step(S1{}, S0{}, kc + 1);
    }
  } else {
    gload(S0{}, 0);
    lstore(S0{}, 0);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
      if (DBG < 2 && kc + 1 < nk) gload(S0{}, kc + 1);  // in flight while this chunk is multiplied
      if (DBG != 1) multiply(DBG == 4 ? 0 : (kc & 1), [](int) {});
      if (kc + 1 < nk) {
        if (DBG < 2) lstore(S0{}, (kc + 1) & 1);
        if (DBG < 3) __syncthreads();
      }
    }
  }
#pragma unroll
  for (int h = 0; h < MI / 2; ++h)
    batch_epilogue<METRIC, DENSE, MODE == 1>(a, reinterpret_cast<f32x16(&)[2][2]>(acc[2 * h]), s_thr, s_qsq, qbase,
                                             nbase, wm * PM + 64 * h, wn * 64, lane);
}

// ---------------------------------------------------------------------------
// block-wide exact k-th smallest (as an order-preserving u32 key) of `n` float
// keys in global memory; NaN entries (dead rows) never count.  256 threads.
// Same idea as K2: thread minima -> one wave bisects them -> short list of
// everything <= that bound -> exact k-th.  Returns KEY_NAN when fewer than k
// entries are finite (caller then keeps everything).
constexpr int BS_THREADS = 256;       // workgroup size for large batches
constexpr int BS_THREADS_WIDE = 1024;  // ... and for small ones: few workgroups in flight, so each gets more waves
constexpr int BS_LIST = 4096;

constexpr int BS_GROUPS = 2048;  // strided groups (BS_GROUPS / THREADS per thread): the bound stays tight up to k = 1024

struct KthScratch {
  uint32_t lm[BS_GROUPS];
  uint32_t list[BS_LIST];
  uint32_t n_list, U, tau;
  RadixSelScratch rs;
};

__device__ __forceinline__ uint32_t fkey_or_dead(float f) { return f != f ? KEY_DEAD : f2key(f); }

// fn(value, index) over keys[0, n), a workgroup's threads striding over 16-byte vectors (n4 = n / 4 when keys is
// 16-byte aligned, else 0), EIGHT loads in flight per thread: with an atomic append in fn the compiler issues one
// load per iteration and waits for it (batch_sample_select_kernel over 33 k-entry sample rows: 100 -> 91 us).
template <int THREADS, typename F>
__device__ __forceinline__ void for_each_key(const float *keys, int n, int n4, int tid, F fn) {
  const f32x4 *keys4 = reinterpret_cast<const f32x4 *>(keys);
  constexpr int U = 8;
  int i = tid;
  for (; i + (U - 1) * THREADS < n4; i += U * THREADS) {
    f32x4 f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) f[u] = keys4[i + u * THREADS];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) fn(f[u][e], 4 * (i + u * THREADS) + e);
  }
  for (; i < n4; i += THREADS) {
    const f32x4 f = keys4[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) fn(f[e], 4 * i + e);
  }
  for (int j = 4 * n4 + tid; j < n; j += THREADS) fn(keys[j], j);
}

// What the selection sees of entry i: the stored key, or (fp16 keys of an L2 / inner-product index, round 5) the key
// plus a per-entry width -- the UPPER side of a key whose error band depends on its row, key + alpha_q |v|.
struct KeyPlain {
  __device__ __forceinline__ float operator()(float f, int) const { return f; }
};
// (the sample rows' norm bounds, once per call: the sample select looks them up for the keys near its threshold)
static __global__ void __launch_bounds__(256) sample_norms_kernel(const float *sqnorm, float *wn, int32_t n) {
  const int32_t i = (int32_t)(blockIdx.x * 256 + threadIdx.x);
  if (i < n) wn[i] = batch_norm_up(sqnorm[i]);
}
struct KeyPlusRowNorm {  // entry i of a candidate list: its row is rows[i]
  const float *sq;
  const uint32_t *rows;
  float alpha2;
  __device__ __forceinline__ float operator()(float f, int i) const { return f + alpha2 * batch_norm_up(sq[rows[i]]); }
};

// k_pick <= k: which order statistic to return -- the k_pick-th smallest -- once the k smallest are known to be in
// the list (the sample select's estimated threshold; everybody else passes k)
template <int THREADS, typename XF = KeyPlain>
__device__ uint32_t block_kth_of_floats(const float *keys, int n, uint32_t k, KthScratch *sc, uint32_t k_pick,
                                        const XF xf = XF()) {
  constexpr int BS_GPT = BS_GROUPS / THREADS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {
    sc->n_list = 0;
    sc->U = KEY_DEAD;
    sc->tau = KEY_NAN;
  }
  // 2048 strided groups (8 per thread): k distinct groups hold an entry <= the k-th
  // smallest group minimum, so that minimum bounds the k-th smallest entry; about
  // G ln(G / (G - k)) entries lie at or below it (1420 at k = 1024)
  // 16-byte loads when the list is 16-byte aligned (the dense sample rows and the candidate lists are): four
  // consecutive entries then share a group, which is as good a partition as any
  const bool vec = (reinterpret_cast<uintptr_t>(keys) & 15) == 0;
  const int n4 = vec ? n >> 2 : 0;
  const f32x4 *keys4 = reinterpret_cast<const f32x4 *>(keys);
  // a list that fits the LDS list whole (B2's candidate lists, small samples) needs no bound first: no first pass
  // (These kernels live on occupancy -- 70 registers, six workgroups per CU, every phase a chain of dependent
  // round trips: keeping the keys in registers through block_kth_radix, or filling the list by index with eight loads
  // in flight, took them to 114-132 registers and from 71 / 55 us to 100 / 65 us at 1024 queries.)
  const bool narrow = (uint32_t)n >= k && k <= (uint32_t)BS_GROUPS / 2 && n > BS_LIST;
  if (narrow) {  // (workgroup-uniform)
    uint32_t lmin[BS_GPT];
#pragma unroll
    for (int u = 0; u < BS_GPT; ++u) lmin[u] = KEY_DEAD;
    for (int base = 0; base < n4; base += BS_GROUPS) {
#pragma unroll
      for (int u = 0; u < BS_GPT; ++u) {
        int i = base + u * THREADS + tid;
        if (i < n4) {
          const f32x4 f = keys4[i];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            uint32_t x = fkey_or_dead(xf(f[e], 4 * i + e));
            lmin[u] = x < lmin[u] ? x : lmin[u];
          }
        }
      }
    }
    for (int base = 4 * n4; base < n; base += BS_GROUPS) {
#pragma unroll
      for (int u = 0; u < BS_GPT; ++u) {
        int i = base + u * THREADS + tid;
        if (i < n) {
          uint32_t x = fkey_or_dead(xf(keys[i], i));
          lmin[u] = x < lmin[u] ? x : lmin[u];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < BS_GPT; ++u) sc->lm[u * THREADS + tid] = lmin[u];
    __syncthreads();
    const uint32_t Ub = block_kth_radix<THREADS>(sc->lm, (uint32_t)BS_GROUPS, k, &sc->rs);
    if (tid == 0) sc->U = Ub;
  }
  __syncthreads();
  const uint32_t U = sc->U;
  for_each_key<THREADS>(keys, n, n4, tid, [&](float f, int i) {
    uint32_t x = fkey_or_dead(xf(f, i));
    if (x <= U && x != KEY_DEAD) {
      uint32_t p = atomicAdd(&sc->n_list, 1u);
      if (p < BS_LIST) sc->list[p] = x;
    }
  });
  __syncthreads();
  {
    const uint32_t m = sc->n_list;  // (workgroup-uniform)
    uint32_t tau;
    if (m < k) {
      tau = KEY_NAN;  // fewer than k live entries
    } else if (m > BS_LIST) {
      tau = U == KEY_DEAD ? KEY_NAN : U;  // flooded by ties: U is a valid, looser bound
    } else {
      tau = block_kth_radix<THREADS>(sc->list, m, k_pick, &sc->rs);
    }
    __syncthreads();
    if (tid == 0) sc->tau = tau;
  }
  __syncthreads();
  return sc->tau;
}

// absolute band of the batched (MFMA) key: band = tau + 2*delta, rounded up
__device__ __forceinline__ float band_float(uint32_t tau_key, float delta2) {
  uint32_t b = band_of(tau_key, 0.f, delta2);
  if (b >= KEY_NAN) return __builtin_inff();
  return key2f(b);
}

struct SampleSelArgs {
  const float *dense;    // nq_pad x dense_ld
  const float *delta2;   // per query: 2 * error bound of the key
  float *thr;            // out: per-query filter threshold (key space)
  uint32_t *tau_est;     // out: the sample key the threshold was derived from (key encoding; KEY_NAN = no threshold)
  uint32_t *cand_key, *cand_row, *cand_cnt;
  int64_t dense_ld;
  int32_t n_sample, k, cand_cap, row0;
  int32_t k_est;         // which order statistic of the sample the threshold comes from (<= k; k = the proven bound)
  // fp16 keys of an L2 / inner-product index (nullable: every key has the one band delta2 / 2): the band of the key of
  // (query q, row v) is alpha[q] |v| + delta2[q] / 2 + chain2 |thr'_q| / 2, thr' the threshold the filtered pass will
  // start its accumulators from (the chain's partial sums carry it: batch_delta2)
  const float *alpha;    // per query
  const float *sqnorm;   // per row (the shard's |v|^2)
  const float *wnorm;    // per sample row: an upper bound of its norm (sample_norms_kernel)
  float chain2;          // 2 c / (1 - 2 c), rounded up (c: the chain's roundings times the unit roundoff)
  float norm_max;        // the shard's longest row
  // Round 6: the HUB rows' keys (nullable) -- the few thousand rows whose norm alone makes them near EVERY query (L2: the
  // shortest rows, inner product: the longest), scored densely beside the sample.  Their k-th smallest key bounds the
  // k-th smallest key overall by construction (k rows lie at or below it), like the sample's own with k_est = k; the
  // threshold is the smaller of the two.  On a corpus whose neighbours ARE its short rows the sample's estimate lets
  // ~480 rows per query through, all of them on the same few thousand rows -- the key kernel's epilogue then walks
  // crowded tiles register by register --; the hub bound lets through what the final list needs.
  const float *hub_dense;  // nq_pad x hub_ld
  int64_t hub_ld;
  int32_t hub_n;
  const uint32_t *row_ids;  // nullable: sample position row0 + i holds row row_ids[row0 + i] (BatchArgs::row_ids)
};

// B0s: one workgroup per query.  thr[q] = band(tau), tau = the k_est-th smallest sample key; the sample rows at
// or below it open the query's candidate list.
// k_est = k: tau bounds the k-th smallest key of the WHOLE row set by construction (k sample rows lie at or below it).
// k_est < k: tau is an ESTIMATE of that bound, verified after the fact.  The sample is 1 / R of the rows, so with
// k_est = k the filtered pass lets ~k R rows per query through (3200 at k = 100, R = 32), and every survivor costs the
// key kernel's epilogue an append -- two thirds of its epilogue time.  The k smallest keys of the whole set fall
// into the sample Binomial(k, 1 / R) at a time (3.1 on average): the p-th smallest SAMPLE key is below the k-th
// smallest key overall only if p or more of them did, which for p = 14 happens once in a million queries -- and
// then B2 sees it (its list holds fewer than k keys at or below tau) and hands the query to the single-query path.
// Survivors per query: ~p R = 450.
// (A variant that keeps the query's sample keys in registers and so reads them once instead of three times was
// measured: 1024 queries 88 us against 71 -- at 181 registers only two workgroups share a CU and their serial phases
// no longer hide behind each other; 16 queries 17.7 against 20.3 us.  Not kept.)
// thr' = b0 / (1 - 2 c), rounded up: the threshold that still covers b0 once the band grows by c |thr'| on either side
__device__ __forceinline__ float widen_by_chain(float b0, float chain2) {
  if (!(chain2 > 0.f) || !(b0 < __builtin_inff())) return b0;
  const double w = (double)b0 + fabs((double)b0) * (double)chain2;
  float f = (float)w;
  if ((double)f < w) {  // next float up
    uint32_t b = __float_as_uint(f);
    if ((b & 0x7FFFFFFFu) == 0u) b = 1u;
    else if (b & 0x80000000u) b -= 1u;
    else b += 1u;
    f = __uint_as_float(b);
  }
  return f;
}

template <int THREADS, bool ROWW>
__device__ __forceinline__ void batch_sample_select_body(const SampleSelArgs &a) {
  __shared__ KthScratch sc;
  __shared__ uint32_t s_cnt;
  const int q = blockIdx.x, tid = threadIdx.x;
  const float *keys = a.dense + (int64_t)q * a.dense_ld;
  const float *wn = ROWW ? static_cast<const float *>(__builtin_assume_aligned(a.wnorm, 16)) : nullptr;
  const float al = ROWW ? a.alpha[q] : 0.f;
  // tau: the k_est-th smallest sample key -- with per-row bands the dense pass wrote the keys' UPPER sides, key +
  // alpha |v| (a square root, a load and an fma per key and pass in HERE cost a 1024-query call 46-60 us)
  uint32_t tau = block_kth_of_floats<THREADS>(keys, a.n_sample, (uint32_t)a.k, &sc, (uint32_t)a.k_est);
  // B2 forms the same upper sides as (key - width) + 2 width, a rounding or two away from key + width: the statistic
  // it is held to sits a few ulps of the operands above what was found here (far inside the band's own slack)
  if (ROWW && tau < KEY_NAN) {
    const float t = key2f(tau);
    if (t < __builtin_inff()) tau = f2key(widen_by_chain(t + (__builtin_fabsf(t) + al * a.norm_max) * 1e-6f, 1e-6f));
  }
  float thr = band_float(tau, a.delta2[q]);
  if (ROWW) thr = widen_by_chain(thr, a.chain2);
  if (a.hub_dense) {  // (workgroup-uniform) the hub rows' bound: proven (k of them lie at or below it), often far tighter
    __syncthreads();
    uint32_t tau2 = block_kth_of_floats<THREADS>(a.hub_dense + (int64_t)q * a.hub_ld, a.hub_n, (uint32_t)a.k, &sc, (uint32_t)a.k);
    if (ROWW && tau2 < KEY_NAN) {
      const float t = key2f(tau2);
      if (t < __builtin_inff()) tau2 = f2key(widen_by_chain(t + (__builtin_fabsf(t) + al * a.norm_max) * 1e-6f, 1e-6f));
    }
    float thr2 = band_float(tau2, a.delta2[q]);
    if (ROWW) thr2 = widen_by_chain(thr2, a.chain2);
    if (thr2 < thr) {
      thr = thr2;
      tau = tau2;
    }
  }
  if (tid == 0) {
    a.thr[q] = thr;
    a.tau_est[q] = thr == __builtin_inff() ? KEY_NAN : tau;  // (no threshold: nothing to verify)
    s_cnt = 0;
  }
  __syncthreads();
  const int n4 = (reinterpret_cast<uintptr_t>(keys) & 15) == 0 ? a.n_sample >> 2 : 0;
  // (the lists hold the keys' LOWER sides, key - alpha |v| = upper side - 2 alpha |v|: what the filtered pass stores
  // for its survivors.  The row's norm is only looked up for the few keys that could pass at all.)
  const float al2 = 2.f * al, thr_any = ROWW ? thr + al2 * a.norm_max * 1.000001f : thr;
  for_each_key<THREADS>(keys, a.n_sample, n4, tid, [&](float f, int i) {
    if (f <= thr_any) {  // NaN (dead) never passes
      const float x = ROWW ? f - al2 * wn[i] : f;
      if (x <= thr) {
        uint32_t p = atomicAdd(&s_cnt, 1u);
        if (p < (uint32_t)a.cand_cap) {
          a.cand_key[(int64_t)q * a.cand_cap + p] = __float_as_uint(x);
          a.cand_row[(int64_t)q * a.cand_cap + p] = a.row_ids ? a.row_ids[a.row0 + i] : (uint32_t)(a.row0 + i);
        }
      }
    }
  });
  __syncthreads();
  if (tid == 0) a.cand_cnt[(int64_t)q * CC_STRIDE] = s_cnt;
}
template <int THREADS>
__global__ void __launch_bounds__(THREADS) batch_sample_select_kernel(SampleSelArgs a) {
  batch_sample_select_body<THREADS, false>(a);
}
template <int THREADS>
__global__ void __launch_bounds__(THREADS) batch_sample_select_roww_kernel(SampleSelArgs a) {
  batch_sample_select_body<THREADS, true>(a);
}

struct FinalSelArgs {
  const uint32_t *cand_key, *cand_row, *cand_cnt;
  const float *delta2;
  const uint32_t *tau_est;  // per query: the sample key the filter threshold came from (KEY_NAN: everything passed)
  uint8_t *blocks;       // nq blocks (BlockHeader + entries), device
  uint8_t *blocks_host;  // nullable: the same blocks in pinned host memory (headers are stored to both)
  uint32_t *final_rows;  // nq x entries
  int64_t block_bytes;
  int64_t row_base, shard_rows;
  int32_t k, cand_cap, entries, metric;
  // per-row bands (see SampleSelArgs; nullable): the list keys are LOWER sides, key - alpha |v|
  const float *alpha, *sqnorm, *thr, *kmax;
  float chain2;
};

// B2: one workgroup per query: exact k-th smallest key of the candidate list,
// widened by the band; survivors are the rows the f64 rerank will score.
template <int THREADS, bool ROWW>
__device__ __forceinline__ void batch_final_select_body(const FinalSelArgs &a) {
  __shared__ KthScratch sc;
  __shared__ uint32_t s_cnt;
  const int q = blockIdx.x, tid = threadIdx.x;
  const uint32_t total = a.cand_cnt[(int64_t)q * CC_STRIDE];
  const bool over_in = total > (uint32_t)a.cand_cap;
  const int n = (int)(over_in ? (uint32_t)a.cand_cap : total);
  const float *keys = reinterpret_cast<const float *>(a.cand_key + (int64_t)q * a.cand_cap);
  const uint32_t *rows = a.cand_row + (int64_t)q * a.cand_cap;
  // Per-row bands: the list holds LOWER sides x = key - alpha |v|; a row can be among the true k nearest only if its
  // lower side is at or below the k-th smallest UPPER side (x + 2 alpha |v|, + 2 beta for both), which bounds the k-th
  // smallest exact key from above.  With one band for all keys (alpha = 0) that is the old rule, tau + 2 delta.
  uint32_t tau;
  float d2 = a.delta2[q];
  if (ROWW) {
    tau = block_kth_of_floats<THREADS>(keys, n, (uint32_t)a.k, &sc, (uint32_t)a.k,
                                       KeyPlusRowNorm{a.sqnorm, rows, 2.f * a.alpha[q]});
    const float thr_c = __builtin_fminf(a.thr[q], a.kmax[q]);  // the filtered pass's start threshold (capped)
    if (thr_c < __builtin_inff()) d2 = d2 + __builtin_fabsf(thr_c) * a.chain2 * 1.000001f;
  } else {
    tau = block_kth_of_floats<THREADS>(keys, n, (uint32_t)a.k, &sc, (uint32_t)a.k);
  }
  float band = band_float(tau, d2);
  // The list holds every row whose (lower) key is <= tau_est + band width.  It holds the whole top k only if the k-th
  // smallest (upper) key overall is <= tau_est, i.e. if at least k of its entries are: tau (their k-th smallest) <=
  // tau_est.  An estimated threshold (SampleSelArgs::k_est < k) that turns out too tight fails exactly this test.
  const uint32_t te = a.tau_est[q];
  const bool unverified = te < KEY_NAN && (tau >= KEY_NAN || tau > te);
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  // (the candidate lists start at multiples of cand_cap entries: 16-byte aligned)
  const int n4 = (reinterpret_cast<uintptr_t>(keys) & 15) == 0 ? n >> 2 : 0;
  for_each_key<THREADS>(keys, n, n4, tid, [&](float f, int i) {
    if (f <= band) {
      uint32_t p = atomicAdd(&s_cnt, 1u);
      if (p < (uint32_t)a.entries) a.final_rows[(int64_t)q * a.entries + p] = rows[i];
    }
  });
  __syncthreads();
  if (tid == 0) {
    BlockHeader hv;
    const bool over = over_in || unverified || s_cnt > (uint32_t)a.entries;
    hv.count = over ? 0u : s_cnt;
    hv.entries = (uint32_t)a.entries;
    hv.tau_key = tau;
    hv.band_key = f2key(band);
    hv.tiles_hit = total;
    hv.flags = (over ? FLAG_LIST_OVERFLOW : 0u) | (unverified ? FLAG_TAU_UNVERIFIED : 0u);
    hv.k = (uint32_t)a.k;
    hv.metric = (uint32_t)a.metric;
    hv.row_base = a.row_base;
    hv.shard_rows = a.shard_rows;
    hv.pad[0] = hv.pad[1] = hv.pad[2] = hv.pad[3] = 0u;
    *reinterpret_cast<BlockHeader *>(a.blocks + (int64_t)q * a.block_bytes) = hv;
    if (a.blocks_host) *reinterpret_cast<BlockHeader *>(a.blocks_host + (int64_t)q * a.block_bytes) = hv;
  }
}
template <int THREADS>
__global__ void __launch_bounds__(THREADS) batch_final_select_kernel(FinalSelArgs a) {
  batch_final_select_body<THREADS, false>(a);
}
template <int THREADS>
__global__ void __launch_bounds__(THREADS) batch_final_select_roww_kernel(FinalSelArgs a) {
  batch_final_select_body<THREADS, true>(a);
}

// K4 over a batch: blockIdx.y = query, blockIdx.x strides over its candidates.
// Identical arithmetic to rerank_kernel (tsh_kernels.hip.h).
struct RerankBatchArgs {
  const float *rows;
  const float *Q;  // nq_pad x ld
  const uint32_t *final_rows;  // nq x entries
  uint8_t *blocks;      // headers (device)
  uint8_t *out_blocks;  // where the entries go: the device blocks, or their pinned host twin (zero-copy results)
  int64_t block_bytes;
  int64_t ld, row_base;
  int32_t dim, entries, metric;
  int32_t q0;  // first query of this launch (the tail of a batch runs in chunks of queries)
};

// One wave = up to 64 candidates of ONE query, one candidate per lane.  The exact sums must
// add their terms strictly in dimension order -- a serial chain per candidate -- so 64 chains
// run side by side and nothing crosses lanes: each lane streams its own row (16 B at a time,
// 8 loads in flight, the next RB_CH dimensions prefetched while the current ones are added) and
// the query values are wave-uniform (scalar loads).
constexpr int RB_CH = 64;  // dimensions per step (two register buffers of RB_CH floats per lane)

// rp: this lane's row (any valid row for idle lanes), qp: the query (wave-uniform); ld: floats per row, multiple of
// 4; rows and queries are zero beyond dim.  Identical arithmetic to rerank_kernel (tsh_kernels.hip.h).
__device__ __forceinline__ void rerank_lane_sums(const float *__restrict__ rp, const float *__restrict__ qp, int dim,
                                                 int ld, int metric, double *out_s0, double *out_s1) {
#pragma clang fp contract(off)
  double s0 = 0.0, s1 = 0.0;
  f32x4 cur[RB_CH / 4], nxt[RB_CH / 4];
  // No branch around any load (offsets past the row are clamped to its last 16 bytes and never used) and none in
  // the loop body: the compiler then counts the loads in flight (vmcnt(8) before a step's arithmetic) instead of
  // draining the step just prefetched as well.
  auto fetch = [&](f32x4 (&dst)[RB_CH / 4], int base) {
#pragma unroll
    for (int j = 0; j < RB_CH / 4; ++j) {
      const int o = base + 4 * j < ld - 4 ? base + 4 * j : ld - 4;  // through L1: a lane's 8 loads share one 128-B line
      dst[j] = *reinterpret_cast<const f32x4 *>(rp + o);
    }
  };
  const int nfull = dim / RB_CH;  // whole steps; the remaining dim % RB_CH elements follow
  auto run = [&](auto METRIC) {  // the metric is a compile-time constant inside: no per-element branches
    constexpr int M = decltype(METRIC)::value;
    auto term = [&](float qf, float bf) {
      const double qd = (double)qf, bd = (double)bf;
      if (M == METRIC_L2) {
        const double diff = qd - bd;
        s0 = s0 + diff * diff;
      } else {
        s0 = s0 + qd * bd;
        if (M == METRIC_COS) s1 = s1 + bd * bd;
      }
    };
    fetch(cur, 0);
    for (int it = 0; it < nfull; ++it) {
      const int base = it * RB_CH;
      fetch(nxt, base + RB_CH);
      __builtin_amdgcn_sched_barrier(0);  // the prefetch stays AHEAD of the arithmetic (the scheduler sinks it)
#pragma unroll
      for (int j = 0; j < RB_CH / 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) term(qp[base + 4 * j + e], cur[j][e]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < RB_CH / 4; ++j) cur[j] = nxt[j];
    }
    const int base = nfull * RB_CH, m = dim - base;  // wave-uniform
#pragma unroll
    for (int j = 0; j < RB_CH / 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * j + e < m) term(qp[base + 4 * j + e], cur[j][e]);
  };
  if (metric == METRIC_L2) run(std::integral_constant<int, METRIC_L2>{});
  else if (metric == METRIC_COS) run(std::integral_constant<int, METRIC_COS>{});
  else run(std::integral_constant<int, METRIC_IP>{});
  *out_s0 = s0;
  *out_s1 = metric == METRIC_COS ? s1 : 0.0;
}

// K4 over a batch.  Every WAVE is on its own: it owns up to 64 candidates of one query (lane = candidate; the chains
// are serial per candidate, so nothing crosses lanes), brings their rows in pieces of 64 floats -- sixteen 1 KiB loads
// (four rows x 256 B each) in flight while it walks the chains of the previous piece --, turns "lane = column" into
// "lane = candidate" through a 17 KB LDS tile of its own, and takes the query from scalar registers.  No workgroup
// barrier anywhere: with eight such waves per CU the load and chain phases of different waves interleave by themselves.
// (Round 2's shape -- a workgroup of four loading waves around one chain wave, 1 KiB row pieces, the query as f64
// in LDS -- had three of four waves idle through every chain phase and all workgroups of a launch in the same phase at
// the same time: 262 us for the 130 k candidates of a 1024-query C3 call against 215 us here in the same four
// launches.  A wave here is latency-bound -- 12 pieces, one prefetched, ~4 us each -- so what matters is how many
// waves are in flight: all queries of a call in ONE launch, rerank_final_kernel below.)
//
// Arithmetic, bit for bit that of rerank_kernel / the oracle (vs_exact_sums): every term is added in dimension
// order with one rounding per addition.  For IP and cosine the product of two f32 values is exact in f64 (24 + 24
// significand bits), so fma(q, b, s) rounds exactly what s + q * b rounds; L2's (q - b)^2 is not exact and keeps its
// separate multiply.
// Shape: a wave owns CAND = 64 candidates and keeps DEPTH = 1 piece in flight beside the one it works on, two waves per
// workgroup (a 1024-query C3 call: 2048 such waves, eight per CU).  Measured against it on one box: 16 candidates per
// wave with six pieces in flight, eight waves per workgroup (64 queries, 8 k candidates: 73 us against 67; no difference
// from 16 to 256 queries per call) -- the waves do not wait for their loads (s_memtime stamps: < 300 cycles per piece),
// a piece costs ~4.5 k cycles of which the chains are 3.2 k.
constexpr int RW_P = 64;               // floats of a row per piece
constexpr int RW_LD = RW_P + 4;        // LDS row stride in floats: rows 4 banks apart
constexpr int RW_CAND = 128;           // candidates per workgroup and pass, either shape
// (128-float pieces measured with the BIG shape, 1024-query C3 call: 180 us against 126 -- four waves per CU instead of
// eight; two pieces in flight per BIG wave: 133; plain instead of nontemporal loads: 136.)
template <int CAND, int DEPTH, int P = RW_P, int NWAVES = RW_CAND / CAND>
struct RwShape {
  static constexpr int WAVES = NWAVES;            // waves per workgroup
  static constexpr int NDEPTH = DEPTH;            // pieces in flight beside the one being worked on
  static constexpr int LD = P + 4;                // LDS row stride in floats: rows 4 banks apart
  static constexpr int TILE = CAND * LD;          // floats of a wave's LDS tile
  static constexpr int LPR = P / 4;               // lanes per row of a 1 KiB load (16 B each)
  static constexpr int RPL = 64 / LPR;            // rows a load covers: 4 x 256 B at P = 64, 8 x 128 B at P = 32
  static constexpr int NLOAD = CAND / RPL;        // 1 KiB load instructions per piece
};
using RwBig = RwShape<64, 1>;
// The finalising kernel's shape (round 4): FOUR waves of 64 candidates per query and 32-float pieces.  Lists are ~130
// entries (k = 100): on two waves that is three wave passes, two of them in sequence, the last for a handful of
// candidates at the price of a full one (a pass is latency: twelve pieces of dependent f64 chains).  Four waves take a
// list of up to 256 in ONE round; with 32-float pieces a wave's tile is 9.2 KB instead of 17.4, so four such
// workgroups still share a CU and all 1024 queries of a call are resident at once.  Measured on the 1024-query C3
// call, same box, alternating (tools/r4_rf_ab.sh): 133 -> 106-107 us; two pieces in flight (DEPTH 2): 110; 16-float pieces:
// 154.  Lists of ~240 (the bench's L2 corpus) take 189 us either way: four wave passes are two rounds on two waves and
// still one round of longer pieces here.
using RwFin = RwShape<64, 1, 32, 4>;

// One wave: the exact sums of candidates [c0, c0 + min(left, CAND)) of query q's list, lane = candidate (lanes past
// the list, or past CAND, repeat the last candidate).  tile: this wave's TILE floats of LDS.
template <int CAND, int DEPTH, int P = RW_P, int NWAVES = RW_CAND / CAND>
__device__ __forceinline__ void rerank_wave_sums(const RerankBatchArgs &a, int q, uint32_t c0, uint32_t left, float *tile,
                                                 int lane, uint32_t *out_row, double *out_s0, double *out_s1) {
#pragma clang fp contract(off)
  using T = RwShape<CAND, DEPTH, P, NWAVES>;
  constexpr int RW_LD = T::LD;  // (shadows the namespace constant: this shape's own stride)
  const uint32_t have = left < (uint32_t)CAND ? left : (uint32_t)CAND;
  const uint32_t slot = (uint32_t)lane < have ? (uint32_t)lane : have - 1u;
  const uint32_t my_row = a.final_rows[(int64_t)q * a.entries + c0 + slot];
  const int ld = (int)a.ld, dim = a.dim;
  const int sub = lane / T::LPR, col4 = 4 * (lane % T::LPR);
  uint32_t rrow[T::NLOAD];  // load j of a piece covers candidates RPL j .. RPL j + RPL - 1, P floats each
#pragma unroll
  for (int j = 0; j < T::NLOAD; ++j) rrow[j] = (uint32_t)__shfl((int)my_row, T::RPL * j + sub);
  const float *__restrict__ qp = a.Q + (int64_t)q * a.ld;  // wave-uniform: scalar loads
  const float *trow = tile + (lane < CAND ? lane : CAND - 1) * RW_LD;
  float *tput = tile + sub * RW_LD + col4;
  const int npiece = (dim + P - 1) / P;
  f32x4 in[DEPTH][T::NLOAD];
  auto fetch = [&](f32x4 (&dst)[T::NLOAD], int p) {  // no branch around a load: offsets past the row clamp to its last 16 bytes (never used)
    const int o = p * P + col4 < ld - 4 ? p * P + col4 : ld - 4;
#pragma unroll
    for (int j = 0; j < T::NLOAD; ++j)
      dst[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(a.rows + (int64_t)rrow[j] * a.ld + o));
  };
  double s0 = 0.0, s1 = 0.0;
  auto chains = [&](auto METRIC) {
    constexpr int M = decltype(METRIC)::value;
    auto term = [&](float qf, float bf) {
      const double qd = (double)qf, bd = (double)bf;
      if (M == METRIC_L2) {
        const double diff = qd - bd;
        s0 = s0 + diff * diff;
      } else {
        s0 = __builtin_fma(qd, bd, s0);
        if (M == METRIC_COS) s1 = __builtin_fma(bd, bd, s1);
      }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) fetch(in[d], d);  // (past the last piece: clamped re-reads of cached lines)
    for (int p0 = 0; p0 < npiece; p0 += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int p = p0 + d;
        if (p < npiece) {  // wave-uniform
#pragma unroll
          for (int j = 0; j < T::NLOAD; ++j) *reinterpret_cast<f32x4 *>(tput + T::RPL * j * RW_LD) = in[d][j];
          fetch(in[d], p + DEPTH);
          __builtin_amdgcn_sched_barrier(0);  // the prefetch is issued BEFORE the chains, not sunk below them
          const float *qq = qp + p * P;
          const int m = dim - p * P;  // wave-uniform; >= 1
          if (m >= P) {
#pragma unroll
            for (int g = 0; g < P / 4; ++g) {
              const f32x4 v = *reinterpret_cast<const f32x4 *>(trow + 4 * g);
#pragma unroll
              for (int e = 0; e < 4; ++e) term(qq[4 * g + e], v[e]);
            }
          } else {
            for (int i = 0; i < m; ++i) term(qq[i], trow[i]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };
  if (a.metric == METRIC_L2) chains(std::integral_constant<int, METRIC_L2>{});
  else if (a.metric == METRIC_COS) chains(std::integral_constant<int, METRIC_COS>{});
  else chains(std::integral_constant<int, METRIC_IP>{});
  *out_row = my_row;
  *out_s0 = s0;
  *out_s1 = a.metric == METRIC_COS ? s1 : 0.0;
}

// Entries out: blockIdx.y = query, blockIdx.x strides over its candidates.  (Shard mode, quarantined rows, wide
// lists: whoever merges or finalises on the host wants the exact sums.)
static __global__ void __launch_bounds__(64 * RwBig::WAVES) rerank_batch_kernel(RerankBatchArgs a) {
  using T = RwBig;
  __shared__ __attribute__((aligned(16))) float tiles[T::WAVES][T::TILE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = a.q0 + (int)blockIdx.y;
  const uint8_t *blk = a.blocks + (int64_t)q * a.block_bytes;
  uint32_t count = reinterpret_cast<const BlockHeader *>(blk)->count;
  if (count > (uint32_t)a.entries) count = (uint32_t)a.entries;
  const uint32_t c0 = (blockIdx.x * (uint32_t)T::WAVES + (uint32_t)wave) * 64u;
  if (c0 >= count) return;
  const uint32_t left = count - c0;  // >= 1
  float *tile = tiles[wave];
  uint32_t my_row;
  double s0, s1;
  rerank_wave_sums<64, 1>(a, q, c0, left, tile, lane, &my_row, &s0, &s1);
  // entries: 24 bytes each, 64 of them contiguous -- staged in the wave's tile and stored as whole dwords side by side
  uint32_t *stage = reinterpret_cast<uint32_t *>(tile);
  {
    const int64_t id = a.row_base + (int64_t)my_row;
    stage[6 * lane + 0] = (uint32_t)id;
    stage[6 * lane + 1] = (uint32_t)((uint64_t)id >> 32);
    stage[6 * lane + 2] = (uint32_t)__double2loint(s0);
    stage[6 * lane + 3] = (uint32_t)__double2hiint(s0);
    stage[6 * lane + 4] = (uint32_t)__double2loint(s1);
    stage[6 * lane + 5] = (uint32_t)__double2hiint(s1);
  }
  asm volatile("" ::: "memory");  // (one wave, LDS in program order: only the compiler must not reorder)
  static_assert(sizeof(BlockEntry) == 24, "entry layout");
  uint32_t *outw = reinterpret_cast<uint32_t *>(a.out_blocks + (int64_t)q * a.block_bytes + sizeof(BlockHeader)) + 6 * (int64_t)c0;
  const uint32_t nw = 6u * (left < 64u ? left : 64u);
#pragma unroll
  for (int i = 0; i < 6; ++i)
    if ((uint32_t)(lane + 64 * i) < nw) outw[lane + 64 * i] = stage[lane + 64 * i];
}

// K4 + the finaliser in one launch: one workgroup per query re-ranks ALL its candidates and then does what the host
// finaliser does (tsh_lib.hip finalize_query; ref ngh_graph_engine.dart:908-946,127,133-134): the distance from the
// exact sums -- L2 sqrt(s0), IP -s0, cosine 1 - s0 / (sqrt(mag_a) * sqrt(s1)) with similarity 0 when the denominator
// is not positive; f64 sqrt and divide are correctly rounded on the device like on the host, contraction off --,
// the strict `> threshold` drop, double.compareTo order as integer keys with the id as tie-break, the cut to k.  The
// k results per query (ids, distances, count) are stored straight into pinned host memory: 1.6 KB per query cross
// PCIe instead of a 3 KB candidate list, and the host is left with a copy.
// Order = rank by counting over the <= RF_MAX candidates in LDS (a list is k plus a band's worth of rows).
constexpr int RF_MAX = 512;                  // candidates per query this kernel takes (wider lists: host finaliser)

struct RerankFinalArgs {
  RerankBatchArgs r;
  const double *sqrt_mag_a;  // per query: sqrt of the query's sum of squares (cosine; element order, f64)
  double thr;                // distance threshold (NaN: none)
  int64_t *out_ids;          // nq x k
  double *out_dist;          // nq x k
  int32_t *out_count;        // nq
  uint32_t *out_info;        // nq: the list's header in one word, flags << 24 | count (the host needs nothing else of it)
  int32_t k;
};

__device__ __forceinline__ uint64_t order_key_of(double d) {  // tsh_lib.hip dart_order_key
  if (d != d) return ~0ull;
  const uint64_t b = (uint64_t)__double_as_longlong(d);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double order_key_to_double(uint64_t key) {
  if (key == ~0ull) return __builtin_nan("");
  return __longlong_as_double((long long)((key >> 63) ? (key & 0x7FFFFFFFFFFFFFFFull) : ~key));
}

template <typename T>
static __global__ void __launch_bounds__(64 * T::WAVES) rerank_final_kernel(RerankFinalArgs fa) {
#pragma clang fp contract(off)
  constexpr int CAND = 64;
  constexpr int RF_MAXG = RF_MAX / (T::WAVES * CAND);  // rounds of WAVES x 64 candidates per workgroup
  static_assert(RF_MAXG * T::WAVES * CAND == RF_MAX, "whole rounds");
  __shared__ __attribute__((aligned(16))) float tiles[T::WAVES][T::TILE];
  __shared__ uint32_t s_valid;
  const RerankBatchArgs &a = fa.r;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = a.q0 + (int)blockIdx.x;
  const BlockHeader *hd = reinterpret_cast<const BlockHeader *>(a.blocks + (int64_t)q * a.block_bytes);
  uint32_t count = hd->count;  // 0 for a list that overflowed or failed its check: the host redoes that query
  if (count > (uint32_t)a.entries) count = (uint32_t)a.entries;
  if (count > (uint32_t)RF_MAX) count = 0;  // (the host does not launch this kernel for such lists)
  const bool has_thr = fa.thr == fa.thr;
  const double sqrt_mag_a = a.metric == METRIC_COS ? fa.sqrt_mag_a[q] : 0.0;
  uint64_t key[RF_MAXG];
  uint32_t row[RF_MAXG];
#pragma unroll
  for (int g = 0; g < RF_MAXG; ++g) {
    key[g] = 0;
    row[g] = 0xFFFFFFFFu;  // = no entry (past the list, a lane past CAND, or beyond the threshold)
    const uint32_t c0 = (uint32_t)(g * T::WAVES + wave) * (uint32_t)CAND;
    if (c0 < count) {  // wave-uniform
      uint32_t r;
      double s0, s1;
      rerank_wave_sums<CAND, T::NDEPTH, T::LD - 4, T::WAVES>(a, q, c0, count - c0, tiles[wave], lane, &r, &s0, &s1);
      double d;
      if (a.metric == METRIC_L2) {
        d = __builtin_sqrt(s0);
      } else if (a.metric == METRIC_IP) {
        d = -s0;
      } else {
        const double denom = sqrt_mag_a * __builtin_sqrt(s1);
        const double sim = denom > 0 ? s0 / denom : 0;
        d = 1.0 - sim;
      }
      if (lane < CAND && (uint32_t)lane < count - c0 && !(has_thr && d > fa.thr)) {
        key[g] = order_key_of(d);
        row[g] = r;
      }
    }
  }
  if (tid == 0) s_valid = 0;
  __syncthreads();  // every tile is free: the lists take their place
  uint64_t *skey = reinterpret_cast<uint64_t *>(&tiles[0][0]);        // RF_MAX keys
  uint32_t *srow = reinterpret_cast<uint32_t *>(skey + RF_MAX);       // RF_MAX local rows
  static_assert(RF_MAX * 12 <= T::WAVES * T::TILE * 4, "lists fit the tiles");
  uint32_t mine = 0;
#pragma unroll
  for (int g = 0; g < RF_MAXG; ++g) {
    const uint32_t c = (uint32_t)(g * T::WAVES + wave) * (uint32_t)CAND + (uint32_t)lane;
    if (lane < CAND && c < count) {
      skey[c] = key[g];
      srow[c] = row[g];
      mine += row[g] != 0xFFFFFFFFu;
    }
  }
  if (mine) atomicAdd(&s_valid, mine);
  __syncthreads();
  const uint32_t n_valid = s_valid;
  const uint32_t kk = (uint32_t)fa.k, n_out = n_valid < kk ? n_valid : kk;
  int64_t *oid = fa.out_ids + (int64_t)q * fa.k;
  double *odist = fa.out_dist + (int64_t)q * fa.k;
#pragma unroll
  for (int g = 0; g < RF_MAXG; ++g) {
    if (row[g] == 0xFFFFFFFFu) continue;
    const uint64_t mk = key[g];
    const uint32_t mr = row[g];
    uint32_t rank = 0;
    // (eight entries per turn, their LDS reads issued together: one entry per turn waited out an LDS round trip per
    // entry -- 12 us of a 64-query call's 67)
#pragma unroll 8
    for (uint32_t j = 0; j < count; ++j) {
      const uint64_t ok = skey[j];
      const uint32_t orow = srow[j];
      rank += (orow != 0xFFFFFFFFu) & ((ok < mk) | ((ok == mk) & (orow < mr)));
    }
    if (rank < kk) {
      oid[rank] = a.row_base + (int64_t)mr;
      odist[rank] = order_key_to_double(mk);
    }
  }
  for (uint32_t i = n_out + (uint32_t)tid; i < kk; i += 64 * T::WAVES) {  // unused slots read as "no row"
    oid[i] = -1;
    odist[i] = __builtin_nan("");
  }
  if (tid == 0) {
    fa.out_count[q] = (int32_t)n_out;
    fa.out_info[q] = (hd->flags << 24) | (hd->count & 0xFFFFFFu);
  }
}

}  // namespace tsh
