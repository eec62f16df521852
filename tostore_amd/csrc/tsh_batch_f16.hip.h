// tsh_batch_f16.hip.h -- the fp16 pre-filter key kernel of the batched path (TSH_OPT_BATCH_KERNEL = 2 / auto),
// second generation.  Same contract as batch_score_bf16x3_kernel<MODE = 1> in tsh_batch.hip.h (operands rounded to
// fp16 after an exact power-of-two scaling, ONE v_mfma_f32_32x32x16_f16 per product, key transform + per-query
// threshold filter fused into the epilogue); different machine mapping:
//
//   * workgroup = 8 waves (2 x 4), each a 128 x 64 patch (4 x 2 MFMA 32x32 blocks, 128 accumulator registers) of a
//     256 x 256 (queries x rows) tile -- 0.75 KB of LDS reads per MFMA instead of 1 KB (first generation: sixteen
//     64 x 64 waves) -- or, for calls of up to 128 queries, a 64 x 64 patch of a 128 x 256 tile.  Every SIMD holds
//     two waves of the workgroup.  Waves 0-3 -- the older wave of each SIMD, which wins the matrix pipe and would
//     otherwise sit at the barrier waiting for its partner -- also issue ALL the loads, one 8 KB block of the stage
//     each, one 1 KB piece behind every pair of their MFMAs; waves 4-7 only read fragments and multiply.
//   * operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write pass.
//     The DMA writes a wave's 64 x 16 B linearly, so the bank swizzle the fragment reads need is baked into the
//     GLOBAL layout of the planes ("plane32" below): a tile's K-chunk is a contiguous run of 8 KB blocks that are
//     byte for byte the LDS image.
//   * K-chunk = 32 (64 B per row), ring of four 32 KB stages, ONE s_barrier per K-step: a wave waits for its own
//     DMA pieces of chunk kc with a COUNTED s_waitcnt vmcnt (the pieces of chunk kc + 1 stay in flight across the
//     barrier), passes the barrier, issues the pieces of chunk kc + 2 into the stage chunk kc - 1 has just left,
//     and multiplies chunk kc.  The DMA is issued from inline asm, so the compiler's own s_waitcnt insertion does
//     not see it (it would drain it before every LDS read); the counts are kept by hand.
//   * survivors of the filter are collected in a per-wave LDS list without atomics (slot = running scalar count +
//     mbcnt of the hit mask) and flushed once per tile: one returning global atomic per survivor, all in flight
//     together, instead of one wait per hit site.
//
// Plane layout "plane32": rows in groups of 128; per group and K-chunk of 32 one 8 KB block;
//   16-byte piece(row, kc, p) at  (((row / 128) * KC + kc) * 128 + row % 128) * 4 + (p ^ ((row >> 2) & 3))
// where piece p holds k = 32 kc + 8 p .. + 7 as fp16.  ds_read_b128 of a fragment (lane = row % 32, piece fixed)
// then touches 16 distinct 16-byte bank groups per 16-lane service group (MI355X_MICROARCH.md, LDS table).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tsh_batch.hip.h"

namespace tsh {

constexpr int F16_KC = 32;        // k per chunk
constexpr int F16_GROUP = 128;    // rows per plane block
constexpr int F16_BLOCK_PIECES = F16_GROUP * 4;  // 16-byte pieces per (group, chunk) block = 8 KB

__host__ __device__ __forceinline__ int64_t plane32_piece(int64_t row, int kc, int p, int kchunks) {
  return (((row / F16_GROUP) * kchunks + kc) * F16_GROUP + row % F16_GROUP) * 4 + (p ^ (int)((row >> 2) & 3));
}

struct Half32Args {
  const float *rows;      // n x ld f32
  const float *inv_norm;  // nullable: multiply each row by its 1/|row| first (cosine corpus)
  u32x4 *out;             // plane32 layout
  int64_t ld;
  int64_t first, n;
  int32_t dim, kchunks;   // kchunks = ceil(dim / 32)
  float scale;            // power of two
  const uint32_t *ids;    // nullable: a GATHERED copy -- plane row first + i holds row ids[i] of `rows` (the hub rows)
};

// one thread = 8 consecutive k of one row = one 16-byte piece
static __global__ void __launch_bounds__(256) half_rows32_kernel(Half32Args a) {
  const int64_t per_row = (int64_t)a.kchunks * 4;
  const int64_t total = a.n * per_row, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t row = a.first + i / per_row;
    const int pp = (int)(i % per_row), kc = pp >> 2, p = pp & 3;
    const int k0 = kc * 32 + p * 8;
    const int64_t srow = a.ids ? (int64_t)a.ids[i / per_row] : row;
    const float *src = a.rows + srow * a.ld + k0;
    const float mul = a.inv_norm ? a.inv_norm[srow] * a.scale : a.scale;
    f16x8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = (_Float16)(k0 + e < a.dim ? src[e] * mul : 0.f);  // round to nearest even
    a.out[plane32_piece(row, kc, p, a.kchunks)] = __builtin_bit_cast(u32x4, h);
  }
}

// ---- the norm-grouped plane (round 6) ---------------------------------------------------------------------------------
// Under L2 / inner product with varying norms a query's survivors are not spread over the corpus: the SHORT rows (L2;
// the long ones under inner product) are near every query, a few thousand rows take most of the ~480 survivors of each
// of a call's 1024 queries, and in the key kernel's epilogue a lane (= one corpus row) that holds more than two
// survivors sends its whole wave tile through the register-by-register walk.  With those rows scattered over the
// corpus a quarter of all wave tiles hold one (the L2 key passes ran 11-14 % behind cosine's, same MFMA work); put
// side by side they fill one wave tile in a hundred.  So the fp16 plane of an L2 / inner-product shard holds its rows
// in another order than the row store: inside every block of PG_ROWS consecutive rows, by norm.  A block maps onto
// itself -- any window of whole blocks (the sample window, a mask's window) holds the same rows in either order, so
// the sample stays an unbiased one and the masks' kept-row counts stay what they were -- and nothing downstream of a
// candidate's row id changes: perm[position] = row travels with the plane, the key kernel reads norms by position
// (psq) and live / mask bits and candidate ids through perm (BatchArgs::row_ids).  The sorted order is ROTATED by a
// block-dependent number of 256-row tiles: with every block's hottest tile at the same offset the persistent key
// kernel's workgroup b (tiles b, b + G, ...; G = 256) would meet either all of them or none (measured on a pre-sorted
// corpus: 8 of 256 workgroups held every crowded tile, 2.2 ms instead of 1.5).  Rows of norm zero (absent /
// quarantined: never live) go last.  The tail behind the last whole block keeps the row order.
constexpr int PG_ROWS = 8192;
struct PlaneGroupArgs {
  const float *sqnorm;    // |v|^2 per row
  uint32_t *perm;         // out, per plane position: the row it holds
  float *psq;             // out, per plane position: that row's |v|^2
  int64_t first;          // first position to (re)write: a multiple of PG_ROWS
  int64_t rows;           // the shard's rows: positions [first, rows) are written
  int32_t longest_first;  // inner product: the longest rows lead a block (L2: the shortest)
};
// one workgroup per block; ranks by counting (a row's rank = the rows of its block in front of it by (norm, place)): 67 M
// compares per block, all blocks side by side -- about a millisecond, once per (re)build of a block
static __global__ void __launch_bounds__(1024) plane_group_kernel(PlaneGroupArgs a) {
  __shared__ __attribute__((aligned(16))) uint32_t s_key[PG_ROWS];
  constexpr int PER = PG_ROWS / 1024;
  const int tid = threadIdx.x;
  const int64_t b0 = a.first + (int64_t)blockIdx.x * PG_ROWS;
  if (b0 + PG_ROWS > a.rows) {  // the tail: row order
    for (int64_t p = b0 + tid; p < a.rows; p += 1024) {
      a.perm[p] = (uint32_t)p;
      a.psq[p] = a.sqnorm[p];
    }
    return;
  }
  // the order key: the norm's leading 19 bits (sign, exponent, 10 of the mantissa: grouping needs no more) above the row's
  // place in the block -- distinct per row, so the ranks are a permutation, and one 32-bit compare per pair
  static_assert(PG_ROWS == 8192, "13 bits of the order key are the row's place in its block");
  for (int i = tid; i < PG_ROWS; i += 1024) {
    const uint32_t u = __float_as_uint(a.sqnorm[b0 + i]);  // (a sum of squares: no sign bit, the bit patterns order like the values)
    const uint32_t kk = (u == 0u || u >= 0x7F800000u) ? 0xFFFFFFFFu : (a.longest_first ? 0x7F800000u - u : u);
    s_key[i] = (kk & 0xFFFFE000u) | (uint32_t)i;
  }
  __syncthreads();
  uint32_t own[PER], rank[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    own[u] = s_key[tid + 1024 * u];
    rank[u] = 0;
  }
  for (int j = 0; j < PG_ROWS; j += 4) {
    const u32x4 k4 = *reinterpret_cast<const u32x4 *>(&s_key[j]);  // (wave-uniform address: one broadcast read)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int u = 0; u < PER; ++u) rank[u] += k4[e] < own[u] ? 1u : 0u;
    }
  }
  const uint32_t rot = 256u * (uint32_t)(((b0 / PG_ROWS) >> 1) & (PG_ROWS / 256 - 1));
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int64_t row = b0 + tid + 1024 * u, pos = b0 + (int64_t)((rank[u] + rot) & (uint32_t)(PG_ROWS - 1));
    a.perm[pos] = (uint32_t)row;
    a.psq[pos] = a.sqnorm[row];
  }
}

// out[i] = src[ids[i]] (the norms of a gathered copy's rows, by position)
static __global__ void __launch_bounds__(256) gather_f32_kernel(const float *src, const uint32_t *ids, float *out, int32_t n) {
  const int32_t i = (int32_t)(blockIdx.x * 256 + threadIdx.x);
  if (i < n) out[i] = src[ids[i]];
}

// one wave's 64 x 16 B from global (per-lane address) into LDS at lds_dst + 16 * lane (wave-uniform base in M0).
// Invisible to the compiler's s_waitcnt bookkeeping: completion is counted by hand (f16_wait_dma).
__device__ __forceinline__ void f16_dma16(const void *gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
// the same with a wave-uniform base in an SGPR pair and one per-lane byte offset shared by all pieces of the wave
__device__ __forceinline__ void f16_dma16_s(const void *sbase_any, uint32_t voff, uint32_t lds_dst) {
  // (the base is wave-uniform by construction; where the compiler has lost track of that it holds it in VGPRs, and the
  // "s" constraint fails to assemble: readfirstlane is free when the value already sits in SGPRs)
  const uint64_t pb = (uint64_t)sbase_any;
  const void *sbase = reinterpret_cast<const void *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pb >> 32)) << 32) |
                                                     (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pb));
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}
template <int N>
__device__ __forceinline__ void f16_wait_dma() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

constexpr int F16_HITS = 128;  // per-wave survivor list of one tile (two per lane); fuller tiles take the slow walk

// PERSISTENT: launch with gridDim.x = min(tiles, CUs) workgroups of 512 threads (one per CU); workgroup b takes tiles
// b, b + gridDim.x, ... of batch_tile_of's XCD-aware order.  The K-chunks of a workgroup's tiles form ONE stream
// through the ring: the first chunks of tile t + 1 are in flight while tile t's epilogue runs, and the global
// atomics that reserve list slots for tile t's survivors are only waited for at the end of tile t + 1.
template <int METRIC, bool DENSE, int MI>
__global__ void __launch_bounds__(512, 2) batch_score_f16_kernel(BatchArgs a) {
  static_assert(MI == 4 || MI == 2, "wave patch 32 MI x 64; 8 waves as 2 x 4");
  constexpr int WN = 4, TM = 64 * MI, TN = 256;
  constexpr int STAGE = (TM + TN) * 64;            // bytes: [queries TM x 64 B][rows TN x 64 B] = 32 / 24 KB
  constexpr int NST = 4, AHEAD = NST - 1;          // ring stages; chunks in flight ahead of the multiply
  constexpr bool IPLIKE = METRIC != METRIC_L2;     // cosine planes hold unit rows: its key is -dot, as for IP
  __shared__ __attribute__((aligned(1024))) unsigned char ring[NST * STAGE];
  __shared__ __attribute__((aligned(16))) float s_thr[TM];
  __shared__ __attribute__((aligned(16))) float s_qsq[TM];
  __shared__ uint2 s_hits[8][F16_HITS + 64];  // + one spare slot per lane

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int KC = a.hchunks;
  const int G = (int)gridDim.x, total_tiles = a.q_tiles * a.n_tiles;

  // ---- issue stream: the chunks of my tiles in order, AHEAD steps in front of the multiply ------------------------
  // The stage image is NBLK = (TM + TN) / 128 blocks of 8 KB, each one contiguous in the planes.  Waves 0 .. NBLK - 1
  // move one block each (eight 1 KB pieces per step: scalar base + lane * 16); waves NBLK .. 7 issue nothing.  The
  // issuing waves are the ones that otherwise wait at the barrier (step timeline, TSH_F16_DBG=32: ~540 of 1800
  // cycles) for partners whose own share of the issue came after their multiply, with the matrix pipe idle.
  // The stream never runs dry: past the last chunk it re-reads the last tile (into stages nobody reads any more),
  // so that the number of pieces in flight -- which the hand-kept vmcnt counts rely on -- is the same at every step.
  constexpr int NBLK = STAGE / 8192;
  const bool issuer = wave < NBLK;
  const unsigned char *blk_src = nullptr;  // wave-uniform: this wave's block of the next chunk of the stream
  int i_tile = (int)blockIdx.x;
  auto set_src = [&](int tile) {
    int qt, nt;
    batch_tile_of(a, tile, &qt, &nt);
    const int qb = qt * TM, nb = a.row0 + nt * TN;  // row0 is a multiple of the tile (host)
    const bool is_q = wave < TM / F16_GROUP;
    const int64_t group = is_q ? (qb / F16_GROUP + wave) : (nb / F16_GROUP + (wave - TM / F16_GROUP));
    const uint64_t p = (uint64_t)(reinterpret_cast<const unsigned char *>(is_q ? a.Qs : a.Vs) + (group * KC) * 8192);
    // (wave-uniform by construction; said explicitly so that the DMA's scalar base operand always gets SGPRs)
    blk_src = reinterpret_cast<const unsigned char *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(p >> 32)) << 32) |
                                                      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)p));
  };
  const uint32_t lane_off = (uint32_t)lane * 16u;
  const uint32_t my_dst = (uint32_t)(uintptr_t)ring + (uint32_t)wave * 8192u;
  // rows past the end of the corpus: the planes are allocated in whole 256-row groups, so the loads stay inside
  // the allocation; what they return is discarded by the epilogue (col_ok)
  auto issue_piece = [&](auto SIDX, int u) {  // 1 KB piece u of this wave's block of the next chunk -> ring stage SIDX
    constexpr int ST = decltype(SIDX)::value;
    f16_dma16_s(blk_src + u * 1024, lane_off, my_dst + (uint32_t)(ST * STAGE + u * 1024));
  };
  auto issue = [&](auto SIDX) {  // the whole block at once (prologue)
    if (issuer) {
#pragma unroll
      for (int u = 0; u < 8; ++u) issue_piece(SIDX, u);
      blk_src += 8192;
    }
  };
  auto next_tile_src = [&]() {  // the stream moves on to my next tile (or stays on the last one)
    if (i_tile + G < total_tiles) i_tile += G;
    set_src(i_tile);
  };
  constexpr int P = 8;  // pieces an issuing wave has in flight per chunk (vmcnt bookkeeping; the others have none)

  // fragment addresses: row = 32 i + (lane & 31) (+ patch base), logical piece 2 s + (lane >> 5), slot = piece ^ sw
  const int half = lane >> 5, l31 = lane & 31, sw = (l31 >> 2) & 3;
  const int a_off = (wm * 32 * MI + l31) * 64, b_off = TM * 64 + (wn * 64 + l31) * 64;
  const int slot0 = ((0 + half) ^ sw) * 16, slot1 = ((2 + half) ^ sw) * 16;
  f32x16 acc[MI][2];
  // Fragments of the two k16 slabs of a chunk, double-buffered ACROSS the barrier: while slab 0 of chunk g is
  // multiplied, slab 1 is read; while slab 1 is multiplied, slab 0 of chunk g + 1 is read -- which is legal because
  // a step's barrier already guarantees chunk g + 1 (see the K loop).  After a barrier the matrix pipe starts at
  // once instead of after all eight waves' LDS reads.
  f16x8 fa[2][MI], fb[2][2];
  auto read_slab = [&](auto SIDX, int sl) {
    constexpr int ST = decltype(SIDX)::value;
    const unsigned char *st = ring + ST * STAGE;
    const int so = sl == 0 ? slot0 : slot1;
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[sl][i] = *reinterpret_cast<const f16x8 *>(st + a_off + i * 2048 + so);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[sl][j] = *reinterpret_cast<const f16x8 *>(st + b_off + j * 2048 + so);
  };
  // One slab's MFMAs; an issuing wave puts one DMA piece of the next chunk behind every pair of them.  (All eight
  // pieces in one go at the top of the step held the LDS / VMEM issue path for ~430 cycles, and the SIMD partner,
  // which has to issue its fragment reads right then, sat blocked in front of its first slab for ~850 cycles:
  // step timeline, TSH_F16_DBG=32.)
  // (Two more wave-uniform tests per pair in here cost 0.16 ms of 1.31; two compile-time copies of the loop, one per
  // role, were no faster than this one test: 1.36 vs 1.31 ms.)
  auto mfma_slab = [&](int sl, auto FILL) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[sl][i], fb[sl][j], acc[i][j], 0, 0, 0);
      if (issuer) {
        constexpr int PER = 8 / (2 * MI);  // pieces behind each pair: 1 (MI = 4) or 2 (MI = 2)
#pragma unroll
        for (int e = 0; e < PER; ++e) issue_piece(FILL, (sl * MI + i) * PER + e);
      }
    }
  };

  // survivors whose list slots have been asked for but not yet answered (two per lane: F16_HITS = 128)
  uint2 pend_e[2] = {uint2{0u, 0u}, uint2{0u, 0u}};
  uint32_t pend_p[2] = {0u, 0u}, pend_n = 0;
  int pend_qbase = 0, pend_nbase = 0;
  auto ask = [&](int qb, int qrow) -> uint32_t {  // reserve one slot in the query's candidate list
    int q = qb + qrow;
    asm volatile("" : "+v"(q));  // keeps the 64 list addresses of a patch from being precomputed (and spilled)
    return atomicAdd(&a.cand_cnt[(int64_t)q * CC_STRIDE], 1u);
  };
  auto store = [&](int qb, int nb, uint32_t e_key, uint32_t e_loc, uint32_t p) {
    if (p < (uint32_t)a.cand_cap) {
      const int64_t o = (int64_t)(qb + (int)(e_loc >> 16)) * a.cand_cap + p;
      a.cand_key[o] = e_key;
      a.cand_row[o] = (uint32_t)(nb + (int)(e_loc & 0xFFFFu));
    }
  };
  auto settle = [&]() {  // the answers have had a whole tile's time to arrive
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if ((uint32_t)(lane + 64 * h) < pend_n) store(pend_qbase, pend_nbase, pend_e[h].x, pend_e[h].y, pend_p[h]);
    pend_n = 0;
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  using S3 = std::integral_constant<int, 3>;
  // probe (TSH_F16_DBG & 32): shader-clock stamps of waves 0 and 4 of workgroup 0 over their first 96 steps
  // Compiled in with -DTSH_PROBES only (tools/f16_probe.sh builds such a library): a release build cannot be
  // switched into the no-epilogue mode, whose results are wrong, by an environment variable.
  int dbg_step = 0;
#ifdef TSH_PROBES
  const bool dbg_on = (a.dbg & 32) && blockIdx.x == 0 && (wave == 0 || wave == 4);
  auto stamp = [&](int point) {
    if (dbg_on && dbg_step < 96 && lane == 0)
      a.dbg_buf[((wave >> 2) * 96 + dbg_step) * 12 + point] = __builtin_amdgcn_s_memtime();
  };
#else
  auto stamp = [](int) {};
#endif
  set_src(i_tile);
  int cur_q_tile = -1;
  // KC is a multiple of four (host) and at least four: every tile starts in ring stage 0
  issue(S0{});
  issue(S1{});
  issue(S2{});
  f16_wait_dma<2 * P>();  // my pieces of chunk 0
  __builtin_amdgcn_s_barrier();
  read_slab(S0{}, 0);

  for (int tile = (int)blockIdx.x; tile < total_tiles; tile += G) {
    int n_tile, q_tile;
    batch_tile_of(a, tile, &q_tile, &n_tile);
    const int qbase = q_tile * TM, nbase = a.row0 + n_tile * TN;
    if (q_tile != cur_q_tile) {
      // Per-query thresholds of this workgroup's query tile: with gridDim.x a multiple of 8 q_tiles a workgroup keeps
      // its query tile for all its tiles, so this runs once (and in the ragged tail of the tile order).  The
      // compiler waits for these loads with vmcnt(0), which also drains the DMA in flight: not in the K loop.
      __syncthreads();  // (nobody still reads the old thresholds)
      for (int t = tid; t < TM; t += 512) {
        const int q = qbase + t;
        float th = -__builtin_inff();  // dense mode / padding rows: nothing passes
        if (!DENSE && q < a.nq) th = a.thr[q];
        if (IPLIKE) th = -th / a.dot_scale;  // key = -(acc * scale) <= thr  <=>  acc >= -thr / scale (power of two: exact)
        s_thr[t] = th;
        s_qsq[t] = METRIC == METRIC_L2 ? a.qsq[q] : 0.f;
      }
      cur_q_tile = q_tile;  // (the K loop's barriers order these stores before the epilogue's reads)
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- K loop: one barrier per step, four steps (one turn of the ring) per iteration ---------------------------
    // On entry to a step: slab 0 of its chunk is in registers; the three chunks after it have been issued.  Stage
    // indices are compile-time constants and the stream's bookkeeping is a pointer increment, because everything a
    // wave does between the barrier and its first MFMA is matrix-pipe idle time: both waves of a SIMD do it at the
    // same moment (a first version with run-time stages and counters spent ~100 instructions there: 67 % MFMA busy
    // with the loads and the epilogue switched off).
    auto step = [&](auto CUR, auto NXT, auto FILL, auto SWITCH) {
      stamp(0);
      f16_wait_dma<P>();             // my pieces of the NEXT chunk have landed (the one after it may still fly)
      stamp(1);
      __builtin_amdgcn_s_barrier();  // everybody's have; and everybody is done reading the chunk before this one
      stamp(2);
      if (decltype(SWITCH)::value) next_tile_src();
      stamp(3);
      // (the sched_barriers pin the order reads -> multiply -> reads -> multiply: left alone the scheduler sinks
      // every read to just before its first use to save registers, and the wave then waits out the LDS latency)
      read_slab(CUR, 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_slab(0, FILL);  // (+ the first half of chunk + 3, into the stage the previous chunk has just left)
      __builtin_amdgcn_sched_barrier(0);
      stamp(7);
      read_slab(NXT, 0);  // legal: this step's barrier already covers the next chunk
      __builtin_amdgcn_sched_barrier(0);
      mfma_slab(1, FILL);
      if (issuer) blk_src += 8192;
      __builtin_amdgcn_sched_barrier(0);
      stamp(4);
      stamp(5);
      ++dbg_step;
    };
    using No = std::false_type;
    using Yes = std::true_type;
    for (int kq = 0; kq < KC / 4 - 1; ++kq) {
      step(S0{}, S1{}, S3{}, No{});
      step(S1{}, S2{}, S0{}, No{});
      step(S2{}, S3{}, S1{}, No{});
      step(S3{}, S0{}, S2{}, No{});
    }
    step(S0{}, S1{}, S3{}, No{});   // issues my last chunk
    step(S1{}, S2{}, S0{}, Yes{});  // ... and from here on the first three chunks of my next tile
    step(S2{}, S3{}, S1{}, No{});
    step(S3{}, S0{}, S2{}, No{});
#ifdef TSH_PROBES
    if (a.dbg & 4) continue;
#endif

    // ---- epilogue ----------------------------------------------------------------------------------------------
    // C/D map of the 32x32 MFMA: col = lane & 31 (corpus row), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (query).
    // The 16 MI (block, register) sites are unrolled (accumulators are registers), so a site's code is kept
    // minimal: compare, ballot, and -- only when a lane of the wave passes -- an append to the wave's LDS list at
    // n_hits + mbcnt(mask); no atomics, nothing to wait for.  At the end every listed survivor asks for its slot
    // (one returning global atomic each, all in flight together); the answers are used one tile later (settle).
    // A tile with more survivors than the list holds (ties, duplicated rows) is walked a second time with one
    // atomic per survivor straight from the registers.
    stamp(6);
    if (!DENSE) settle();  // the previous tile's
    stamp(8);
    uint32_t n_hits = 0;   // wave-uniform
    uint2 *my_hits = s_hits[wave];
    // (opaque per tile: everything the epilogue derives from the lane id is invariant across the tiles of the
    // persistent loop, and the compiler would hoist all 16 MI site constants out of it -- and spill them)
    int half_t = half, l31_t = l31;
    asm volatile("" : "+v"(half_t), "+v"(l31_t));
    // per column block j: the corpus row this lane holds, and whether it may be returned
    int cl[2];
    bool col_ok[2];
    float vsq[2];
    uint64_t alive_m[2];
    bool alive_l[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      cl[j] = wn * 64 + j * 32 + l31_t;  // row within the tile
      const int col = nbase + cl[j];
      col_ok[j] = col < a.row1;
      bool alive = col_ok[j];
      vsq[j] = 0.f;
      if (col_ok[j]) {
        if (METRIC == METRIC_L2) vsq[j] = a.sqnorm[col];
        if (a.live) alive = (a.live[col >> 6] >> (col & 63)) & 1ull;
        if (alive && a.mask) alive = (a.mask[col >> 6] >> (col & 63)) & 1ull;
      }
      alive_l[j] = alive;
      alive_m[j] = __ballot(alive);
    }
    auto walk = [&](auto DIRECT) {
      constexpr bool direct = decltype(DIRECT)::value;
      // (the second walk must not share values with the first: the compiler would keep all of the first walk's
      // 16 MI keys and list entries alive for it -- in scratch)
      float scale_w = a.dot_scale;
      int half_w = half_t;
      asm volatile("" : "+s"(scale_w), "+v"(half_w));
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int rbase = wm * 32 * MI + i * 32 + 4 * half_w;  // tile row (query) of reg 0
        f32x4 th[4], qq[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          th[gq] = *reinterpret_cast<const f32x4 *>(&s_thr[rbase + 8 * gq]);
          if (METRIC == METRIC_L2) qq[gq] = *reinterpret_cast<const f32x4 *>(&s_qsq[rbase + 8 * gq]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {  // the four registers of one group of consecutive query rows
            auto key_of = [&](int r) -> float {
              if (IPLIKE) return -(acc[i][j][r] * scale_w);
              const float dot = acc[i][j][r] * scale_w;
              return qq[r >> 2][r & 3] + vsq[j] - 2.f * dot;
            };
            if (DENSE) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int r = 4 * g4 + e, qrow = rbase + e + 8 * g4;
                if (col_ok[j])
                  a.dense[(int64_t)(qbase + qrow) * a.dense_ld + (nbase + cl[j] - a.row0)] = alive_l[j] ? key_of(r) : __builtin_nanf("");
              }
            } else {
              // Four compares into four scalar masks, ONE test and branch for the four sites; keys, live / mask bits
              // and list slots are only worked out behind the branch, which four groups in nine take (four sites in
              // five have no survivor).  The walk takes ~10 k cycles per tile (epilogue stamps of TSH_F16_DBG=32):
              // 4.3 k for the compares and branches of the 128 sites, 6.4 k for the ~14 visits behind the branch --
              // instruction issue at ~4.5 cycles per instruction and wave with two waves per SIMD, nothing else
              // (one compare + branch per site, inline or out of line, took the same; margins reduced with v_max3
              // in the vector unit plus branch-free appends took 14 k).
              uint64_t m0[4];
              bool pass_l[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int r = 4 * g4 + e;
                pass_l[e] = IPLIKE ? acc[i][j][r] >= th[g4][e] : key_of(r) <= th[g4][e];
                m0[e] = __ballot(pass_l[e]);
              }
              if (__builtin_expect((m0[0] | m0[1] | m0[2] | m0[3]) != 0, 0)) {  // wave-uniform
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const uint64_t m = m0[e] & alive_m[j];
                  if (m) {
                    const int r = 4 * g4 + e, qrow = rbase + e + 8 * g4;
                    const bool mine = pass_l[e] && alive_l[j];  // (a lane mask to the compiler: no bit arithmetic)
                    const float key = key_of(r);
                    if (direct) {
                      if (mine) store(qbase, nbase, __float_as_uint(key), ((uint32_t)qrow << 16) | (uint32_t)cl[j], ask(qbase, qrow));
                    } else {
                      // Kept SMALL (this block exists 128 times per patch and each copy runs once in a few tiles:
                      // its instructions come from L2, not from the instruction cache -- ~460 cycles per visit with
                      // the first, larger form): no exec games, no capacity test; a lane that does not pass, or
                      // whose slot is beyond the list, writes to its own spare slot behind it.
                      const uint32_t slot = n_hits + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                      const uint32_t spare = (uint32_t)F16_HITS + (uint32_t)lane;
                      const uint32_t at = mine ? (slot < spare ? slot : spare) : spare;
                      my_hits[at] = uint2{__float_as_uint(key), ((uint32_t)qrow << 16) | (uint32_t)cl[j]};
                      n_hits += (uint32_t)__popcll(m);
                    }
                  }
                }
              }
            }
          }
        }
      }
    };
    walk(std::false_type{});
    stamp(9);
    if (!DENSE) {
      if (n_hits > (uint32_t)F16_HITS) {
        walk(std::true_type{});
      } else {
#pragma unroll
        for (int h = 0; h < 2; ++h)
          if ((uint32_t)(lane + 64 * h) < n_hits) {
            pend_e[h] = my_hits[lane + 64 * h];
            pend_p[h] = ask(qbase, (int)(pend_e[h].y >> 16));
          }
        pend_n = n_hits;
        pend_qbase = qbase;
        pend_nbase = nbase;
      }
    }
    stamp(10);
  }
  if (!DENSE) settle();
  f16_wait_dma<0>();  // the stream's surplus pieces must have landed before this workgroup's LDS is handed on
}

}  // namespace tsh
