// tsh_kernels.hip.h -- gfx950 (MI355X, CDNA4) device code for the exhaustive
// kNN path behind ToStore's vectorSearch().  Wave = 64 lanes everywhere.
//
// Pipeline for ONE query over a resident shard (rows: n x ld float32, row-major):
//   K1 scan     HBM-bound.  One wave owns a tile of 64 rows; every row is read
//               exactly once with coalesced 16-byte/lane loads (1 KiB per wave
//               instruction), per-lane partial sums are combined with a
//               transposing butterfly so that lane l ends up holding the f32
//               ranking key of row tile*64+l.  Writes keys[n] (4 B/row, 0.13 %
//               of the bytes read at d=768) and gmin[tile] = min key of the tile.
//   K2 select   one workgroup.  tau = k-th smallest of gmin[] is a PROVEN upper
//               bound on the k-th smallest key (k distinct tiles hold a row at
//               or below it).  tau is widened by the f32 error band, then only
//               the tiles with gmin <= band are re-read from keys[] and rows at
//               or below the band become candidates (~k of them).
//   K3 filter   fallback when K2's candidate list overflows (ties / degenerate
//               data): whole-grid filter of keys[] against the band.
//   K4 rerank   exact f64 re-accumulation, element order 0..d-1, one rounding
//               per multiply and per add (no FMA), of the reference's
//               _l2Distance / _innerProduct / _cosineSimlarity
//               (/root/reference/lib/src/core/ngh_graph_engine.dart:920-946)
//               for the candidates only.  sqrt / divide / sort stay on the host.
//
// Keys order like the reference's distances: L2 -> sum of squares (sqrt is
// monotone), IP -> -dot, cosine -> -dot/|v| (query norm is a common factor).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tsh {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr uint32_t KEY_DEAD = 0xFFFFFFFFu;  // masked / deleted / absent row
constexpr uint32_t KEY_NAN = 0xFFFFFFFEu;   // live row whose key is NaN (sorts last)
constexpr int METRIC_L2 = 0, METRIC_IP = 1, METRIC_COS = 2;

// order-preserving float -> uint32 (smaller float <=> smaller key; -0 < +0)
__device__ __forceinline__ uint32_t f2key(float f) {
  if (f != f) return KEY_NAN;
  uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
  uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return __uint_as_float(b);
}

// ---------------------------------------------------------------------------
// transposing butterfly: x[0..7] per lane, lane bits P..P+2.  Returns, in each
// lane, sum over the 8 lanes that differ from it only in bits P..P+2 of
// x[idx], idx = (lane >> P) & 7.  7 exchanges instead of 8*3.  The pairing tree
// is the same for every idx and float add commutes, so a row's result does not
// depend on its position in the tile.
template <int P>
__device__ __forceinline__ float treduce8(const float (&x)[8], int lane) {
  float y[4], z[2];
  bool b0 = (lane >> P) & 1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float keep = b0 ? x[2 * i + 1] : x[2 * i];
    float send = b0 ? x[2 * i] : x[2 * i + 1];
    y[i] = keep + __shfl_xor(send, 1 << P);
  }
  bool b1 = (lane >> (P + 1)) & 1;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float keep = b1 ? y[2 * i + 1] : y[2 * i];
    float send = b1 ? y[2 * i] : y[2 * i + 1];
    z[i] = keep + __shfl_xor(send, 1 << (P + 1));
  }
  bool b2 = (lane >> (P + 2)) & 1;
  float keep = b2 ? z[1] : z[0];
  float send = b2 ? z[0] : z[1];
  return keep + __shfl_xor(send, 1 << (P + 2));
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    uint32_t t = (uint32_t)__shfl_xor((int)v, o);
    v = t < v ? t : v;
  }
  return v;
}

template <int METRIC>
__device__ __forceinline__ float accum4(float acc, f32x4 q, f32x4 v) {
  if (METRIC == METRIC_L2) {
    float d0 = q.x - v.x, d1 = q.y - v.y, d2 = q.z - v.z, d3 = q.w - v.w;
    acc = __builtin_fmaf(d0, d0, acc);
    acc = __builtin_fmaf(d1, d1, acc);
    acc = __builtin_fmaf(d2, d2, acc);
    acc = __builtin_fmaf(d3, d3, acc);
  } else {
    acc = __builtin_fmaf(q.x, v.x, acc);
    acc = __builtin_fmaf(q.y, v.y, acc);
    acc = __builtin_fmaf(q.z, v.z, acc);
    acc = __builtin_fmaf(q.w, v.w, acc);
  }
  return acc;
}

template <bool NT>
__device__ __forceinline__ f32x4 ld16(const float *p) {
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
  return *reinterpret_cast<const f32x4 *>(p);
}

struct ScanArgs {
  const float *rows;      // n x ld
  const float *query;     // ld floats, zero padded past dim; NULL: the query rides in
                          // the kernel-argument segment (ScanArgsQ::q), no H2D copy needed
  float *query_out;       // nullable: workgroup 0 stores the query here for the rerank kernel
  const float *inv_norm;  // cosine: 1/|row| (0 for zero rows), else unused
  const uint64_t *live;   // masked variant: bit r of word t = row t*64+r present & not deleted
  const uint64_t *mask;   // masked variant, nullable: caller's keep mask, same layout
  uint32_t *keys;         // n_tiles*64
  uint32_t *gmin;         // n_tiles
  int64_t ld;             // floats per row, multiple of 4
  int64_t n;              // rows
  int32_t d4;             // float4 per row (= ld/4)
  int32_t n_tiles;        // ceil(n/64); list scans: ceil(list entries / 64) -- keys / gmin are in LIST order then
  const uint32_t *list;   // list scans (scan_list_kernel): local row ids of the rows to scan, ascending, padded with
                          // 0xFFFFFFFF to a multiple of 64; NULL otherwise
};

// Kernel parameter block: the scan arguments plus up to 960 query floats inline,
// 3936 bytes of the 4 KiB kernel-argument segment.
constexpr int SCAN_Q_INLINE = 960;
struct ScanArgsQ {
  ScanArgs a;
  float q[SCAN_Q_INLINE];
};
static_assert(sizeof(ScanArgsQ) <= 4096, "kernel arguments are limited to 4 KiB");

// K1.  NCH >= ceil(d4/64) 16-byte chunks per lane per row.  FULL: d4 == NCH*64;
// otherwise lanes past the row end re-read their chunk 0 (a cache hit) and the
// value is zeroed, so there is no branch around any load.
// Rows are consumed in groups of R: while group i is being reduced, group i+1
// is already in flight (two register buffers; R*NCH KiB per buffer per wave).
// The compiler-opaque fences pin that schedule so the register budget, and
// with it the wave occupancy, is what the template arguments say.
// "memory" stops IR passes moving loads across; sched_barrier(0) stops the
// machine scheduler moving the (register-only) math across.
#define TSH_FENCE()                       \
  do {                                    \
    asm volatile("" ::: "memory");        \
    __builtin_amdgcn_sched_barrier(0);    \
  } while (0)

template <int NCH, int METRIC, bool FULL, bool MASKED, int R, bool NT, int WAVES, int MINW>
__global__ void __launch_bounds__(WAVES * 64, MINW) scan_kernel(ScanArgsQ aq) {
  static_assert(R == 2 || R == 4, "R must give an even number of groups per 8-row batch");
  const ScanArgs &a = aq.a;
  const float *qsrc = a.query;
  if (!qsrc) {
    typedef const char __attribute__((address_space(4))) * karg_ptr;
    qsrc = (const float *)((karg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(ScanArgsQ, q));
  }
  constexpr int G = 8 / R;  // groups per batch
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // WAVES bounds the workgroup size; small shards are launched with one wave
  // per workgroup so the dispatcher can balance tiles across CUs
  const int wpb = __builtin_amdgcn_readfirstlane((int)blockDim.x >> 6);
  const int stride = gridDim.x * wpb;
  // !FULL: the row ends inside chunk d4/64 and every later chunk is empty (NCH comes from a short list of
  // widths, so more than the last chunk can lie beyond the row).  vmask bit c = this lane's 16 bytes of chunk
  // c exist; lanes without them reload the row's first 16 bytes (valid memory) and contribute zeros.
  uint32_t vmask = 0;
#pragma unroll
  for (int c = 0; c < NCH; ++c) vmask |= (FULL || c * 64 + lane < a.d4) ? (1u << c) : 0u;
  auto has = [&](int c) { return FULL || ((vmask >> c) & 1u) != 0u; };
  // (row and query pointers already include + 4 * lane: lanes without data fall back to element 0 of the row --
  // 4 * lane floats further on may be past the end of the last row's allocation when rows are narrow)
  auto off = [&](int c) { return has(c) ? c * 256 : -4 * lane; };
  uint32_t loff[NCH];  // !FULL: this lane's float offset into a row, per chunk
#pragma unroll
  for (int c = 0; c < NCH; ++c) loff[c] = has(c) ? (uint32_t)(4 * lane + c * 256) : 0u;

  f32x4 q[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    q[c] = *reinterpret_cast<const f32x4 *>(qsrc + 4 * lane + off(c));
    if (!has(c)) q[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (a.query_out && blockIdx.x == 0 && wave == 0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      if (has(c))
        *reinterpret_cast<f32x4 *>(a.query_out + 4 * lane + c * 256) = q[c];
  }

  // masked scans hand consecutive tiles to different WORKGROUPS: a contiguous id-range
  // filter leaves one run of live tiles, which would otherwise land on a few CUs
  for (int t = MASKED ? wave * (int)gridDim.x + (int)blockIdx.x : (int)blockIdx.x * wpb + wave; t < a.n_tiles;
       t += stride) {
    const float *tbase = a.rows + (int64_t)t * 64 * a.ld + 4 * lane;
    uint64_t bits = ~0ull;
    int cnt = 64;
    if (MASKED) {
      uint64_t w = a.live[t];
      if (a.mask) w &= a.mask[t];
      uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)w);
      uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(w >> 32));
      bits = ((uint64_t)hi << 32) | lo;
      cnt = __popcll(bits);
      if (cnt == 0) {
        if (lane == 0) a.gmin[t] = KEY_DEAD;  // keys[] of a dead tile stay stale: every reader checks gmin first
        continue;
      }
    }
    const int nb = MASKED ? (cnt + 7) >> 3 : 8;  // 8-row batches, wave-uniform

    f32x4 v[2][R][NCH];
    uint64_t rem = bits;
    int last = 0, next_dense = 0;
    auto load_group = [&](int buf) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        int r;
        if (MASKED) {  // next live row (a short last batch repeats the final row)
          if (rem) {
            last = __builtin_ctzll(rem);
            rem &= rem - 1;
          }
          r = last;
        } else {
          r = next_dense++;
        }
        if (FULL) {
          const float *rp = tbase + (int64_t)r * a.ld;
#pragma unroll
          for (int c = 0; c < NCH; ++c) v[buf][j][c] = ld16<NT>(rp + off(c));
        } else {
          // a lane's offset differs from chunk to chunk here (lanes past the row's end fall back to its start), which as
          // a 64-bit address per row and chunk cost R x NCH register pairs and spilled (d = 384: 30 registers, d = 1000:
          // 164; 0.62 / 0.42 of the HBM peak where full widths reach 0.82): the row's start is wave-uniform -- a scalar
          // base -- and the lane's part a 32-bit offset per chunk, computed once
          // (through readfirstlane: otherwise the optimiser derives the next row's addresses from this row's, per lane)
          const uint64_t rbi = (uint64_t)(a.rows + ((int64_t)t * 64 + r) * a.ld);
          const float *rb = reinterpret_cast<const float *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(rbi >> 32)) << 32) |
                                                            (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)rbi));
#pragma unroll
          for (int c = 0; c < NCH; ++c) v[buf][j][c] = ld16<NT>(rb + loff[c]);
        }
      }
    };

    float val = 0.f;
    // The previous tile's key / gmin stores share vmcnt with the loads, and stores may retire out of order with
    // loads: while one MIGHT be pending the compiler must wait for vmcnt(0) instead of counting.  Retire them here.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    load_group(0);
    // One 8-row batch.  The LAST batch is a separate instance without the trailing load instead of an `if` inside
    // the loop: behind a branch the compiler's s_waitcnt insertion no longer knows which loads are in flight
    // and waits for vmcnt(0) -- the group just issued included -- before every group's arithmetic.
    auto batch = [&](int b, auto LAST) {
      float acc[8];
#pragma unroll
      for (int k = 0; k < G; ++k) {
        if (!(decltype(LAST)::value && k == G - 1)) load_group((k + 1) & 1);
        TSH_FENCE();
#pragma unroll
        for (int j = 0; j < R; ++j) {
          float s = 0.f;
#pragma unroll
          for (int c = 0; c < NCH; ++c)
            if (has(c)) s = accum4<METRIC>(s, q[c], v[k & 1][j][c]);  // (!FULL: lanes past the row's end sit the chunk out)
          // tie the finished sum to the fence: pure math would otherwise be
          // sunk below the next group's loads, keeping every buffer live
          asm volatile("" : "+v"(s)::"memory");
          acc[k * R + j] = s;
        }
        TSH_FENCE();
      }
      // octet partial of row (lane&7) of this batch, then across the 8 octets
      float o = treduce8<0>(acc, lane);
      o += __shfl_xor(o, 8);
      o += __shfl_xor(o, 16);
      o += __shfl_xor(o, 32);
      if ((lane >> 3) == b) val = o;  // slot b*8 + (lane&7) == lane
    };
#pragma nounroll
    for (int b = 0; b < nb - 1; ++b) batch(b, std::false_type{});
    batch(nb - 1, std::true_type{});
    // dense: val = key sum of row t*64+lane; masked: of the lane-th live row

    bool alive;
    if (MASKED) {
      // expand compact slots back to row positions
      uint64_t below = bits & ((1ull << lane) - 1ull);
      int rank = __popcll(below);
      val = __shfl(val, rank);
      alive = (bits >> lane) & 1ull;
    } else {
      alive = (int64_t)t * 64 + lane < a.n;
    }
    if (METRIC == METRIC_L2) {
      // nothing
    } else if (METRIC == METRIC_IP) {
      val = -val;
    } else {
      float inv = alive ? a.inv_norm[(int64_t)t * 64 + lane] : 0.f;
      val = -(val * inv);
    }
    uint32_t key = alive ? f2key(val) : KEY_DEAD;
    a.keys[(int64_t)t * 64 + lane] = key;
    uint32_t m = wave_min_u32(key);
    if (lane == 0) a.gmin[t] = m;
  }
}

// K1 for SELECTIVE row masks (a WHERE clause that keeps a few percent of the rows; config C5): the host compacts the
// kept row ids into a list once per mask, and the scan is a gather over that list instead of a walk over tiles that
// are mostly dead.  Why: a masked scan_kernel wave meets 0.6 live rows per tile at keep 1 %, reads the tile's mask
// words, then the row, then moves on -- three dependent round trips for 3 KB, every row in another 2 MB page (a TLB
// miss each: profiles/r03_rerank_tlb_counters.txt).  Here ONE wave owns eight consecutive list entries and has all
// eight rows in flight at once; consecutive list entries are neighbours in the row store (the list is ascending),
// so a wave's rows share one or two pages, and all of a mask's waves are resident together.
// A workgroup = 8 waves = 64 list entries = one "tile" of the downstream select kernel, which runs unchanged on the
// compact keys (n_tiles = entries / 64: a 1 % mask of 1 M rows selects among 157 tiles instead of 15 625) and maps
// list positions back to row ids when it emits candidates (SelectArgs::list).
template <int NCH, int METRIC, int R, bool NT>
__global__ void __launch_bounds__(512, 2) scan_list_kernel(ScanArgsQ aq) {
  static_assert(R == 2 || R == 4, "R must give an even number of groups per 8-row batch");
  static_assert(2 * R * NCH * 4 + NCH * 4 <= 200, "two register buffers + the query within 256 VGPRs");
  const ScanArgs &a = aq.a;
  const float *qsrc = a.query;
  if (!qsrc) {
    typedef const char __attribute__((address_space(4))) * karg_ptr;
    qsrc = (const float *)((karg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(ScanArgsQ, q));
  }
  constexpr int G = 8 / R;
  __shared__ uint32_t s_min[8];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t = (int)blockIdx.x;  // one workgroup per 64 list entries
  // this lane's 16 bytes of chunk c exist (the row may end inside a chunk, and NCH comes from a short list of widths)
  uint32_t vmask = 0;
#pragma unroll
  for (int c = 0; c < NCH; ++c) vmask |= (c * 64 + lane < a.d4) ? (1u << c) : 0u;
  auto has = [&](int c) { return ((vmask >> c) & 1u) != 0u; };
  uint32_t loff[NCH];  // lanes past the row's end reload its first 16 bytes (valid memory) and sit the chunk out
#pragma unroll
  for (int c = 0; c < NCH; ++c) loff[c] = has(c) ? (uint32_t)(4 * lane + c * 256) : 0u;
  f32x4 q[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    q[c] = *reinterpret_cast<const f32x4 *>(qsrc + loff[c]);
    if (!has(c)) q[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (a.query_out && blockIdx.x == 0 && wave == 0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      if (has(c)) *reinterpret_cast<f32x4 *>(a.query_out + 4 * lane + c * 256) = q[c];
  }
  // my wave's eight list entries: lanes wave * 8 .. wave * 8 + 7 of the tile (the lanes that end up with their keys)
  const int e0 = wave * 8;
  uint32_t my_id = 0xFFFFFFFFu;
  if ((lane >> 3) == wave) my_id = a.list[(int64_t)t * 64 + lane];
  bool alive = my_id != 0xFFFFFFFFu;
  if (alive) alive = (a.live[my_id >> 6] >> (my_id & 63)) & 1ull;  // (tombstoned since the mask was made: still a dead row)
  const uint64_t valid = __ballot(my_id != 0xFFFFFFFFu) >> e0;  // bit j = entry e0 + j exists (a prefix: padding is last)
  const int cnt = __popcll(valid & 0xFFull);
  float val = 0.f;
  if (cnt > 0) {
    f32x4 v[2][R][NCH];
    auto load_group = [&](int buf, int g) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        int e = g * R + j;
        e = e < cnt ? e : cnt - 1;  // a short batch repeats its last row
        const uint32_t id = (uint32_t)__builtin_amdgcn_readlane((int)my_id, e0 + e);
        const float *rb = a.rows + (int64_t)id * a.ld;  // wave-uniform: a scalar base + one 32-bit lane offset per chunk
#pragma unroll
        for (int c = 0; c < NCH; ++c) v[buf][j][c] = ld16<NT>(rb + loff[c]);
      }
    };
    float acc[8];
    load_group(0, 0);
#pragma unroll
    for (int k = 0; k < G; ++k) {
      if (k + 1 < G) load_group((k + 1) & 1, k + 1);
      TSH_FENCE();
#pragma unroll
      for (int j = 0; j < R; ++j) {
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
          if (has(c)) sum = accum4<METRIC>(sum, q[c], v[k & 1][j][c]);
        asm volatile("" : "+v"(sum)::"memory");
        acc[k * R + j] = sum;
      }
      TSH_FENCE();
    }
    float o = treduce8<0>(acc, lane);  // octet partial of row (lane & 7), then across the 8 octets
    o += __shfl_xor(o, 8);
    o += __shfl_xor(o, 16);
    o += __shfl_xor(o, 32);
    val = o;  // every lane with (lane & 7) == j holds row j's sum: the wave's own lanes e0 + j take theirs
  }
  if (METRIC == METRIC_IP) {
    val = -val;
  } else if (METRIC == METRIC_COS) {
    const float inv = alive ? a.inv_norm[my_id] : 0.f;
    val = -(val * inv);
  }
  const uint32_t key = alive ? f2key(val) : KEY_DEAD;  // (lanes outside the wave's eight: my_id invalid -> KEY_DEAD)
  if ((lane >> 3) == wave) a.keys[(int64_t)t * 64 + lane] = key;
  const uint32_t m = wave_min_u32(key);
  if (lane == 0) s_min[wave] = m;
  __syncthreads();
  if (wave == 0) {
    uint32_t x = lane < 8 ? s_min[lane] : KEY_DEAD;
    x = wave_min_u32(x);
    if (lane == 0) a.gmin[t] = x;
  }
}

// K1 for narrow rows (ld = 128 / 64 / 32 floats): one 1 KiB wave load covers
// 2 / 4 / 8 whole rows, so every lane stays busy.  A "virtual row" is the 256
// floats of one wave load; the butterfly is the same as above but stops before
// its last SPLIT exchange steps, which leaves each of the virtual row's real rows
// in its own group of 64 >> SPLIT lanes.  Masked tiles are not compacted here
// (rows are tiny): a tile is skipped only when none of its 64 rows is live.
template <int SPLIT, int METRIC, bool MASKED, bool NT>
__global__ void __launch_bounds__(256, 4) scan_packed_kernel(ScanArgsQ aq) {
  static_assert(SPLIT >= 1 && SPLIT <= 3, "2, 4 or 8 rows per wave load");
  constexpr int LPR = 64 >> SPLIT;           // lanes per real row
  constexpr int NB = 8 >> SPLIT;             // batches of 8 virtual rows per 64-row tile
  const ScanArgs &a = aq.a;
  const float *qsrc = a.query;
  if (!qsrc) {
    typedef const char __attribute__((address_space(4))) * karg_ptr;
    qsrc = (const float *)((karg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(ScanArgsQ, q));
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpb = __builtin_amdgcn_readfirstlane((int)blockDim.x >> 6);
  const int stride = gridDim.x * wpb;
  const f32x4 q = *reinterpret_cast<const f32x4 *>(qsrc + 4 * (lane & (LPR - 1)));
  if (a.query_out && blockIdx.x == 0 && wave == 0 && lane < LPR)
    *reinterpret_cast<f32x4 *>(a.query_out + 4 * lane) = q;
  // real row (within the tile) whose key this lane ends up holding
  const int my_b = (lane >> 3) & (NB - 1);
  const int rr = (((my_b * 8) + (lane & 7)) << SPLIT) | (lane >> (6 - SPLIT));

  // masked scans hand consecutive tiles to different WORKGROUPS: a contiguous id-range
  // filter leaves one run of live tiles, which would otherwise land on a few CUs
  for (int t = MASKED ? wave * (int)gridDim.x + (int)blockIdx.x : (int)blockIdx.x * wpb + wave; t < a.n_tiles;
       t += stride) {
    const float *tbase = a.rows + (int64_t)t * 64 * a.ld + 4 * lane;
    uint64_t bits = ~0ull;
    if (MASKED) {
      uint64_t w = a.live[t];
      if (a.mask) w &= a.mask[t];
      uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)w);
      uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(w >> 32));
      bits = ((uint64_t)hi << 32) | lo;
      if (bits == 0) {
        if (lane == 0) a.gmin[t] = KEY_DEAD;  // keys[] of a dead tile stay stale: every reader checks gmin first
        continue;
      }
    }
    f32x4 v[2][8];
    auto load_batch = [&](int buf, int b) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[buf][j] = ld16<NT>(tbase + (b * 8 + j) * 256);
    };
    float val = 0.f;
    load_batch(0, 0);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b + 1 < NB) load_batch((b + 1) & 1, b + 1);
      TSH_FENCE();
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float sacc = accum4<METRIC>(0.f, q, v[b & 1][j]);
        asm volatile("" : "+v"(sacc)::"memory");
        acc[j] = sacc;
      }
      float o = treduce8<0>(acc, lane);
      if (LPR >= 16) o += __shfl_xor(o, 8);
      if (LPR >= 32) o += __shfl_xor(o, 16);
      if (my_b == b) val = o;
      TSH_FENCE();
    }
    const int64_t row = (int64_t)t * 64 + rr;
    const bool alive = MASKED ? ((bits >> rr) & 1ull) != 0 : row < a.n;
    if (METRIC == METRIC_IP) {
      val = -val;
    } else if (METRIC == METRIC_COS) {
      float inv = alive ? a.inv_norm[row] : 0.f;
      val = -(val * inv);
    }
    uint32_t key = alive ? f2key(val) : KEY_DEAD;
    a.keys[row] = key;
    uint32_t m = wave_min_u32(key);
    if (lane == 0) a.gmin[t] = m;
  }
}

// ---------------------------------------------------------------------------
// candidate block written by K2/K3/K4 and shipped to the host (and all-gathered
// between ranks): 64-byte header + entries x {int64 id, f64 s0, f64 s1}
struct BlockHeader {
  uint32_t count;      // candidates found (may exceed `entries`: overflow)
  uint32_t entries;    // capacity of this block
  uint32_t tau_key;    // k-th smallest tile minimum
  uint32_t band_key;   // tau widened by the f32 error band
  uint32_t tiles_hit;  // tiles re-read by K2
  uint32_t flags;      // bit0: K2 list overflow (K3 fallback needed)
  uint32_t k;
  uint32_t metric;
  int64_t row_base;
  int64_t shard_rows;
  uint32_t pad[4];     // [0]: a failed rank's status (FLAG_RANK_ERROR); [1]: the block's generation (sharded calls)
};
static_assert(sizeof(BlockHeader) == 64, "header is 64 bytes");
struct BlockEntry {
  int64_t id;
  double s0;  // L2: sum (q-v)^2   IP: sum q*v   cosine: sum q*v
  double s1;  // cosine: sum v*v   else 0
};
static_assert(sizeof(BlockEntry) == 24, "entry is 24 bytes");
constexpr uint32_t FLAG_LIST_OVERFLOW = 1u;
constexpr uint32_t FLAG_TAU_REFINED = 2u;  // informational: select step (f) ran
constexpr uint32_t FLAG_TAU_UNVERIFIED = 4u;  // batched path: the estimated filter threshold was too tight (set with
                                              // FLAG_LIST_OVERFLOW: the query is redone by the single-query path)

struct SelectArgs {
  const uint32_t *gmin;
  const uint32_t *keys;
  BlockHeader *hdr;       // device header (the rerank kernel reads its count)
  BlockHeader *hdr_host;  // nullable: mirror in pinned host memory (zero-copy store)
  uint32_t *cand_rows;  // local row ids, capacity cand_cap
  int32_t n_tiles;
  int32_t k;
  int32_t cand_cap;
  float eps_rel;    // band = tau + |tau|*eps_rel + delta_abs
  float delta_abs;
  int32_t force_all;  // safe mode: every live row is a candidate
  int32_t metric;
  int64_t row_base;
  int64_t shard_rows;
  const uint32_t *list;  // list scans: keys / gmin are in list order; candidate rows = list[position]
  uint32_t tag;          // generation of the block (BlockHeader.pad[1]): 0, or what a sharded call's exchange expects
};

constexpr int SEL_THREADS = 1024;   // one whole CU; tile minima stay in registers
constexpr int SEL_VPT = 16;         // tile minima a thread keeps in registers (16384 tiles = 1M rows)
constexpr int SEL_LIST_CAP = 4096;  // LDS lists (short list of tile minima / tile ids)
constexpr int SEL_PREFILTER_MIN = 512;  // above this many tiles, bound the list by the thread-minima bisect first

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    uint32_t t = (uint32_t)__shfl_xor((int)v, o);
    v = t > v ? t : v;
  }
  return v;
}

// ONE wave, no barriers: smallest X with count(v <= X) >= k over the wave's
// VPT x 64 register values (pad with KEY_DEAD).  Bits above the highest bit in
// which min and max differ are common to every value and are skipped.
template <int VPT>
__device__ uint32_t wave_kth_bisect(const uint32_t (&v)[VPT], uint32_t k) {
  uint32_t lo = KEY_DEAD, hi = 0;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    lo = v[i] < lo ? v[i] : lo;
    hi = v[i] > hi ? v[i] : hi;
  }
  lo = wave_min_u32(lo);
  hi = wave_max_u32(hi);
  uint32_t diff = lo ^ hi;
  if (diff == 0) return lo;
  int top = 31 - __builtin_clz(diff);  // highest differing bit
  uint32_t X = top == 31 ? 0u : (lo >> (top + 1)) << (top + 1);
  for (int bit = top; bit >= 0; --bit) {
    uint32_t T = X | ((1u << bit) - 1u);
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < VPT; ++i) c += (uint32_t)__popcll(__ballot(v[i] <= T));
    if (c < k) X |= (1u << bit);
  }
  return X;
}

// k-th smallest (k = 1 .. m) of m 32-bit keys in LDS, by the WHOLE workgroup: four rounds over one byte of the
// key each, most significant first -- a 256-bin histogram of the keys that still match the prefix found so far
// (LDS atomics, every thread taking its share of the keys), then wave 0 scans the bins and names the one holding
// the k-th key.  wave_kth_bisect above has ONE wave walk up to 32 bits x all keys while the others wait: 13 us
// for 2048 keys and 30 us for a 4096-entry list (measured with s_memtime inside batch_sample_select_kernel, where
// the two of them were half the kernel); this takes 2-3 us whatever the key distribution.
// Every thread of the workgroup must call it (it contains barriers).  rs: scratch in LDS.
struct RadixSelScratch {
  uint32_t hist[256];
  uint32_t bin, k;
};
template <int THREADS>
__device__ uint32_t block_kth_radix(const uint32_t *v, uint32_t m, uint32_t k, RadixSelScratch *rs) {
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 256; i += THREADS) rs->hist[i] = 0u;
  uint32_t prefix = 0u;
  __syncthreads();
#pragma unroll 1
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (uint32_t i0 = 0; i0 < m; i0 += THREADS) {  // (workgroup-uniform trip count: the ballots below see whole waves)
      const uint32_t i = i0 + tid;
      const uint32_t x = i < m ? v[i] : 0u;
      const bool in = i < m && (shift == 24 || (x >> (shift + 8)) == (prefix >> (shift + 8)));
      const uint32_t b = (x >> shift) & 255u;
      // keys that agree in this byte (close keys: the usual case in the upper rounds) would queue up on one bin:
      // a wave whose matching lanes all name the same bin adds its count once
      const uint64_t act = __ballot(in);
      if (act) {
        const uint32_t b0 = (uint32_t)__shfl((int)b, __builtin_ctzll(act));
        if (__ballot(in && b == b0) == act) {
          if (lane == (int)__builtin_ctzll(act)) atomicAdd(&rs->hist[b0], (uint32_t)__popcll(act));
        } else if (in) {
          atomicAdd(&rs->hist[b], 1u);
        }
      }
    }
    __syncthreads();
    if (tid < 64) {  // wave 0: lane l owns bins 4l .. 4l+3 (and clears them for the next round)
      uint32_t c[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        c[j] = rs->hist[4 * lane + j];
        rs->hist[4 * lane + j] = 0u;
      }
      const uint32_t mine = c[0] + c[1] + c[2] + c[3];
      uint32_t incl = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += t;
      }
      uint32_t below = incl - mine;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (k > below && k <= below + c[j]) {  // exactly one (lane, j): 1 <= k <= number of matching keys
          rs->bin = (uint32_t)(4 * lane + j);
          rs->k = k - below;
        }
        below += c[j];
      }
    }
    __syncthreads();
    prefix |= rs->bin << shift;
    k = rs->k;
  }
  __syncthreads();  // (rs may be reused by the caller right away)
  return prefix;
}

// tau widened by the f32 error band, rounded UP to the next float
__device__ __forceinline__ uint32_t band_of(uint32_t tau_key, float eps_rel, float delta_abs) {
  if (tau_key >= KEY_NAN) return KEY_NAN;
  float t = key2f(tau_key);
  if (t == __builtin_inff()) return KEY_NAN;  // overflowed keys: NaN keys may hide finite values
  double w = (double)t + fabs((double)t) * (double)eps_rel + (double)delta_abs;
  float f = (float)w;
  if ((double)f < w) {  // next float up
    uint32_t b = __float_as_uint(f);
    if ((b & 0x7FFFFFFFu) == 0u) b = 1u;           // +-0 -> smallest positive
    else if (b & 0x80000000u) b -= 1u;             // negative: toward zero
    else b += 1u;
    f = __uint_as_float(b);
  }
  if (!(f < __builtin_inff())) return KEY_NAN;
  return f2key(f);
}

// K2.  One workgroup.  Each thread keeps its share of gmin[] in registers
// (IN_REGS: n_tiles <= 16384) so global memory is read once.
//  (a) per-thread minimum -> LDS; wave 0 bisects those 1024 values (or 256
//      4-thread group minima when k <= 128) for U = their k-th smallest:
//      k distinct threads hold a tile minimum <= U, so U >= k-th tile minimum
//  (b) tile minima <= U -> short LDS list (about k(1 + k/2G) entries)
//  (c) wave 0 bisects the short list: tau = exact k-th smallest tile minimum
//  (d) band = tau widened by the f32 error model; tiles with minimum <= band
//  (e) only those tiles' keys are re-read; keys <= band are the candidates
template <int NT, bool IN_REGS>
__device__ __forceinline__ void select_body(const SelectArgs &a) {
  static_assert(NT == 1024 || NT == 256, "thread minima are bisected by one wave as 16 or 4 per lane");
  constexpr int SEL_THREADS = NT;  // shadows the namespace constant inside this kernel
  __shared__ uint32_t s_lm[2 * SEL_THREADS];
  __shared__ uint32_t s_list[SEL_LIST_CAP];
  __shared__ uint32_t s_n, s_tiles, s_cand, s_U, s_tau;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = a.n_tiles;
  const uint32_t k = (uint32_t)a.k;
  if (tid == 0) {
    s_n = 0;
    s_tiles = 0;
    s_cand = 0;
    s_U = KEY_DEAD;
    s_tau = KEY_NAN;
  }
  uint32_t g[SEL_VPT];
  uint32_t lmin = KEY_DEAD, lodd = KEY_DEAD;  // lodd: minimum over this thread's odd-numbered elements
  if (IN_REGS) {
#pragma unroll
    for (int i = 0; i < SEL_VPT; ++i) {
      int t = tid + i * SEL_THREADS;
      g[i] = t < M ? a.gmin[t] : KEY_DEAD;
      lmin = g[i] < lmin ? g[i] : lmin;
      if (i & 1) lodd = g[i] < lodd ? g[i] : lodd;
    }
  } else {
    int it = 0;
    for (int t = tid; t < M; t += SEL_THREADS, ++it) {
      uint32_t x = a.gmin[t];
      lmin = x < lmin ? x : lmin;
      if (it & 1) lodd = x < lodd ? x : lodd;
    }
  }
  const bool narrow = !a.force_all && (uint32_t)M >= k && k <= (uint32_t)SEL_THREADS;
  const bool groups4 = NT == 1024 && k <= 128;  // 256 group minima are plenty for small k
  // above k = 512 the k-th of 1024 thread minima gets loose (G ln(G/(G-k)) entries below
  // it): split every thread into its even and odd elements, 2048 groups
  const bool groups2k = k > 512;
  if (narrow && M > SEL_PREFILTER_MIN) {
    uint32_t m = lmin;
    if (groups4) {
      uint32_t t1 = (uint32_t)__shfl_xor((int)m, 1);
      m = t1 < m ? t1 : m;
      uint32_t t2 = (uint32_t)__shfl_xor((int)m, 2);
      m = t2 < m ? t2 : m;
    }
    s_lm[tid] = m;
    if (groups2k) {  // even-element minimum and odd-element minimum as separate groups
      uint32_t leven = KEY_DEAD;
      if (IN_REGS) {
#pragma unroll
        for (int i = 0; i < SEL_VPT; i += 2) leven = g[i] < leven ? g[i] : leven;
      } else {
        int it = 0;
        for (int t = tid; t < M; t += SEL_THREADS, ++it)
          if (!(it & 1)) {
            uint32_t x = a.gmin[t];
            leven = x < leven ? x : leven;
          }
      }
      s_lm[tid] = leven;
      s_lm[SEL_THREADS + tid] = lodd;
    }
  }
  __syncthreads();
  if (narrow && M > SEL_PREFILTER_MIN && wave == 0) {
    uint32_t U;
    if (groups2k) {
      uint32_t v[2 * NT / 64];
#pragma unroll
      for (int i = 0; i < 2 * NT / 64; ++i) v[i] = s_lm[lane + i * 64];
      U = wave_kth_bisect<2 * NT / 64>(v, k);
    } else if (groups4) {
      uint32_t v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = s_lm[(lane + i * 64) * 4];
      U = wave_kth_bisect<4>(v, k);
    } else {
      uint32_t v[NT / 64];
#pragma unroll
      for (int i = 0; i < NT / 64; ++i) v[i] = s_lm[lane + i * 64];
      U = wave_kth_bisect<NT / 64>(v, k);
    }
    if (lane == 0) s_U = U;
  }
  __syncthreads();
  if (narrow) {
    const uint32_t U = s_U;  // KEY_DEAD when the list holds everything (M <= SEL_PREFILTER_MIN)
    if (IN_REGS) {
#pragma unroll
      for (int i = 0; i < SEL_VPT; ++i) {
        if (tid + i * SEL_THREADS < M && g[i] <= U) {
          uint32_t p = atomicAdd(&s_n, 1u);
          if (p < SEL_LIST_CAP) s_list[p] = g[i];
        }
      }
    } else {
      for (int t = tid; t < M; t += SEL_THREADS) {
        uint32_t x = a.gmin[t];
        if (x <= U) {
          uint32_t p = atomicAdd(&s_n, 1u);
          if (p < SEL_LIST_CAP) s_list[p] = x;
        }
      }
    }
  }
  __syncthreads();
  if (narrow && wave == 0) {
    const uint32_t n = s_n;
    uint32_t tau;
    if (n > SEL_LIST_CAP) {
      tau = s_U;  // ties flooded the list: U is still a valid (looser) bound
    } else if (n <= 512) {
      uint32_t v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (uint32_t)(lane + i * 64) < n ? s_list[lane + i * 64] : KEY_DEAD;
      tau = wave_kth_bisect<8>(v, k);
    } else {
      uint32_t v[SEL_LIST_CAP / 64];
#pragma unroll
      for (int i = 0; i < SEL_LIST_CAP / 64; ++i)
        v[i] = (uint32_t)(lane + i * 64) < n ? s_list[lane + i * 64] : KEY_DEAD;
      tau = wave_kth_bisect<SEL_LIST_CAP / 64>(v, k);
    }
    if (tau == KEY_DEAD) tau = KEY_NAN;  // fewer than k live tiles: every live row
    if (lane == 0) s_tau = tau;
  }
  __syncthreads();
  // A list scan of fewer tiles than k (a selective mask: a few thousand kept rows): the tile minima bound nothing, every
  // tile would be a hit and step (f) would bisect for the k-th key in up to 32 block-wide rounds.  The list-ordered keys
  // are one short contiguous array, every entry written (padding = KEY_DEAD): take the k-th smallest KEY directly,
  // four radix rounds by the whole workgroup.  (79 tiles, k = 100: 60 -> 25 us per query.)
  if (a.list && !a.force_all && (uint32_t)M < k && (uint32_t)M * 64u >= k) {  // workgroup-uniform (fewer entries than k: all)
    __shared__ RadixSelScratch s_rs;
    const uint32_t x = block_kth_radix<NT>(a.keys, (uint32_t)M * 64u, k, &s_rs);
    if (tid == 0) s_tau = x >= KEY_DEAD ? KEY_NAN : x;  // fewer than k live rows: every live row
    __syncthreads();
  }
  const uint32_t tau = s_tau;
  const uint32_t band = a.force_all ? KEY_NAN : band_of(tau, a.eps_rel, a.delta_abs);

  // (d) tiles whose minimum is inside the band (KEY_DEAD > band always)
  if (IN_REGS) {
#pragma unroll
    for (int i = 0; i < SEL_VPT; ++i) {
      if (g[i] <= band) {
        uint32_t p = atomicAdd(&s_tiles, 1u);
        if (p < SEL_LIST_CAP) s_list[p] = (uint32_t)(tid + i * SEL_THREADS);
      }
    }
  } else {
    for (int t = tid; t < M; t += SEL_THREADS) {
      if (a.gmin[t] <= band) {
        uint32_t p = atomicAdd(&s_tiles, 1u);
        if (p < SEL_LIST_CAP) s_list[p] = (uint32_t)t;
      }
    }
  }
  __syncthreads();
  const uint32_t nt = s_tiles;
  const bool over = nt > SEL_LIST_CAP;
  constexpr int NW = SEL_THREADS / 64;
  // (e) four tiles per wave per round so their (L2-resident) key loads overlap
  auto emit = [&](uint32_t bnd) {
    for (uint32_t j0 = wave * 4; j0 < nt; j0 += NW * 4) {
      uint32_t tile[4], key[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        tile[u] = j0 + u < nt ? s_list[j0 + u] : 0xFFFFFFFFu;
        key[u] = tile[u] != 0xFFFFFFFFu ? a.keys[(int64_t)tile[u] * 64 + lane] : KEY_DEAD;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        bool pass = key[u] <= bnd;
        uint64_t bm = __ballot(pass);
        if (bm) {
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(&s_cand, (uint32_t)__popcll(bm));
          base = (uint32_t)__shfl((int)base, 0);
          if (pass) {
            uint32_t p = base + (uint32_t)__popcll(bm & ((1ull << lane) - 1ull));
            if (p < (uint32_t)a.cand_cap) a.cand_rows[p] = a.list ? a.list[tile[u] * 64 + lane] : tile[u] * 64 + lane;
          }
        }
      }
    }
  };
  if (!over) emit(band);
  __syncthreads();
  uint32_t tau_out = tau, band_out = band;
  const uint32_t first_count = s_cand;
  // (f) The k-th smallest TILE MINIMUM is a loose bound when the best rows sit together in a
  // few tiles (rows inserted by topic, a contiguous-range filter, ...): far more than k keys
  // are then <= band and the list overflows.  Every key <= tau lives in a hit tile, and there
  // are at least k of them, so the k-th smallest key over the hit tiles IS the exact k-th
  // smallest key of the shard: bisect for it (block-wide counts; keys in registers when the
  // hit tiles fit, otherwise re-read from L2) and emit again with the tight band.
  // (tau == KEY_NAN, fewer than k live tiles, is covered too: more than cand_cap >= k live keys exist)
  if (!over && !a.force_all && first_count > (uint32_t)a.cand_cap) {
    __shared__ uint32_t s_cnt[3], s_lo;
    constexpr int RV = 16;  // tiles a wave keeps in registers
    const bool in_regs = nt <= (uint32_t)(NW * RV);
    uint32_t kr[RV];
    uint32_t lo = KEY_DEAD;
    if (in_regs) {
#pragma unroll
      for (int i = 0; i < RV; ++i) {
        uint32_t j = (uint32_t)(wave + i * NW);
        kr[i] = j < nt ? a.keys[(int64_t)s_list[j] * 64 + lane] : KEY_DEAD;
        lo = kr[i] < lo ? kr[i] : lo;
      }
    } else {
      for (uint32_t j = wave; j < nt; j += NW) {
        uint32_t x = a.keys[(int64_t)s_list[j] * 64 + lane];
        lo = x < lo ? x : lo;
      }
    }
    if (tid == 0) {
      s_lo = KEY_DEAD;
      s_cnt[0] = s_cnt[1] = s_cnt[2] = 0;
    }
    __syncthreads();  // every thread has read first_count by now
    if (tid == 0) s_cand = 0;
    lo = wave_min_u32(lo);
    if (lane == 0) atomicMin(&s_lo, lo);
    __syncthreads();
    lo = s_lo;
    uint32_t X = lo;
    const uint32_t diff = lo ^ tau;  // lo <= answer <= tau
    if (diff) {
      const int top = 31 - __builtin_clz(diff);
      X = top == 31 ? 0u : (lo >> (top + 1)) << (top + 1);
      int slot = 0;
      for (int bit = top; bit >= 0; --bit) {
        const uint32_t T = X | ((1u << bit) - 1u);
        uint32_t c = 0;
        if (in_regs) {
#pragma unroll
          for (int i = 0; i < RV; ++i) c += (uint32_t)__popcll(__ballot(kr[i] <= T));
        } else {
          for (uint32_t j = wave; j < nt; j += NW)
            c += (uint32_t)__popcll(__ballot(a.keys[(int64_t)s_list[j] * 64 + lane] <= T));
        }
        if (lane == 0) atomicAdd(&s_cnt[slot], c);
        const int nxt = slot == 2 ? 0 : slot + 1;
        if (tid == 0) s_cnt[nxt] = 0;  // nobody touches the next slot before the barrier below
        __syncthreads();
        if (s_cnt[slot] < k) X |= (1u << bit);
        slot = nxt;
      }
    }
    tau_out = X;
    band_out = band_of(X, a.eps_rel, a.delta_abs);
    emit(band_out);
    __syncthreads();
  }
  if (tid == 0) {
    BlockHeader hv;
    hv.count = over ? 0u : s_cand;
    hv.entries = (uint32_t)a.cand_cap;
    hv.tau_key = tau_out;
    hv.band_key = band_out;
    hv.tiles_hit = nt;
    hv.flags = (over || s_cand > (uint32_t)a.cand_cap) ? FLAG_LIST_OVERFLOW : 0u;
    if (tau_out != tau) hv.flags |= FLAG_TAU_REFINED;
    hv.k = (uint32_t)a.k;
    hv.metric = (uint32_t)a.metric;
    hv.row_base = a.row_base;
    hv.shard_rows = a.shard_rows;
    hv.pad[0] = hv.pad[2] = hv.pad[3] = 0u;
    hv.pad[1] = a.tag;
    *a.hdr = hv;
    if (a.hdr_host) *a.hdr_host = hv;
  }
}

template <int NT, bool IN_REGS>
__global__ void __launch_bounds__(NT) select_kernel(SelectArgs a) {
  select_body<NT, IN_REGS>(a);
}

// blocks written by the matrix-core path get their generation afterwards
static __global__ void __launch_bounds__(64) stamp_tag_kernel(uint8_t *blocks, size_t block_bytes, int32_t n, uint32_t tag) {
  const int32_t q = (int32_t)(blockIdx.x * 64 + threadIdx.x);
  if (q < n) reinterpret_cast<BlockHeader *>(blocks + (size_t)q * block_bytes)->pad[1] = tag;
}

// K3a: one pass of a radix select over the WHOLE key array (fallback path, k beyond what the tile-minimum
// bound of K2 can serve): histogram of bits [shift, shift + 8) of the keys that share `prefix` above them.
// Dead tiles are skipped through gmin (their keys are stale); dead rows inside live tiles carry KEY_DEAD.
static __global__ void __launch_bounds__(256) radix_hist_kernel(const uint32_t *keys, const uint32_t *gmin, int64_t n_keys,
                                                         uint32_t prefix, int shift, uint32_t *hist) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_keys; i += stride) {
    if (gmin[i >> 6] == KEY_DEAD) continue;  // a wave covers one tile: uniform
    const uint32_t key = keys[i];
    if (key == KEY_DEAD) continue;
    if (shift < 24 && (key >> (shift + 8)) != (prefix >> (shift + 8))) continue;
    atomicAdd(&h[(key >> shift) & 255u], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// K3: whole-grid filter (fallback).  count accumulates in *out_count.
static __global__ void __launch_bounds__(256) filter_kernel(const uint32_t *keys, const uint32_t *gmin, int64_t n_keys,
                                                     uint32_t band, uint32_t *out_rows,
                                                     uint32_t *out_count, uint32_t cap, const uint32_t *list = nullptr) {
  const int lane = threadIdx.x & 63;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t n_round = (n_keys + 63) & ~(int64_t)63;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
    bool pass = i < n_keys && gmin[i >> 6] != KEY_DEAD && keys[i] <= band;  // a wave covers one tile
    uint64_t bm = __ballot(pass);
    if (bm) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(out_count, (uint32_t)__popcll(bm));
      base = __shfl((int)base, 0);
      if (pass) {
        uint32_t p = base + __popcll(bm & ((1ull << lane) - 1ull));
        if (p < cap) out_rows[p] = list ? list[i] : (uint32_t)i;  // (list scans: keys are in list order)
      }
    }
  }
}

// K4: exact f64 sums for candidate rows.  One wave per candidate: lanes form the
// per-element terms in parallel (each term is one IEEE multiply, so order is
// irrelevant), lane 0 (and lane 1 for the cosine row norm) adds them strictly
// in element order 0..d-1 -- the reference's loop order.
struct RerankArgs {
  const float *rows;
  const float *query;       // ld floats
  const uint32_t *cand_rows;
  const uint32_t *count_ptr;  // device count (clamped to cap)
  BlockEntry *out;
  int64_t ld;
  int64_t row_base;
  int32_t dim;
  int32_t cap;
  int32_t metric;
};
constexpr int RR_CHUNK = 1024;

static __global__ void __launch_bounds__(64) rerank_kernel(RerankArgs a) {
#pragma clang fp contract(off)
  __shared__ __attribute__((aligned(16))) double t0[RR_CHUNK];
  __shared__ __attribute__((aligned(16))) double t1[RR_CHUNK];
  const int lane = threadIdx.x;
  uint32_t count = *a.count_ptr;
  if (count > (uint32_t)a.cap) count = (uint32_t)a.cap;
  const int chains = a.metric == METRIC_COS ? 2 : 1;  // lane 0: s0, lane 1: row norm
  for (uint32_t c = blockIdx.x; c < count; c += gridDim.x) {
    uint32_t row = a.cand_rows[c];
    const float *rp = a.rows + (int64_t)row * a.ld;
    double s = 0.0;
    for (int base = 0; base < a.dim; base += RR_CHUNK) {
      int m = a.dim - base < RR_CHUNK ? a.dim - base : RR_CHUNK;
      for (int i = lane; i < m; i += 64) {
        double qv = (double)a.query[base + i], bv = (double)rp[base + i];
        if (a.metric == METRIC_L2) {
          double diff = qv - bv;
          t0[i] = diff * diff;
        } else {
          t0[i] = qv * bv;
          if (a.metric == METRIC_COS) t1[i] = bv * bv;
        }
      }
      __syncthreads();
      if (lane < chains) {
        // strictly sequential adds; LDS reads are issued 32 elements ahead so
        // only the add chain's own latency remains
        const double *src = lane == 0 ? t0 : t1;
        int i = 0;
        for (; i + 32 <= m; i += 32) {
          double x[32];
#pragma unroll
          for (int u = 0; u < 32; ++u) x[u] = src[i + u];
#pragma unroll
          for (int u = 0; u < 32; ++u) s = s + x[u];
        }
        for (; i < m; ++i) s = s + src[i];
      }
      __syncthreads();
    }
    double s1 = __shfl(s, 1);
    if (lane == 0) {
      a.out[c].id = a.row_base + (int64_t)row;
      a.out[c].s0 = s;
      a.out[c].s1 = a.metric == METRIC_COS ? s1 : 0.0;
    }
  }
}

// Exact sums (rerank_kernel's arithmetic: strictly sequential f64, element order) of a shard's QUARANTINED rows
// -- rows outside the f32 error model, kept out of the scan (see ingest_kernel) -- for nq queries: one lane
// per (row, query).  A handful of rows, so no tiling: each lane walks its own row.
struct QuarArgs {
  const float *rows;
  const float *Q;        // nq queries, ldq floats apart
  const uint32_t *list;  // [0] = count, [1 ..] = local row ids
  BlockEntry *out;       // [query][cap]
  int64_t ld, ldq, row_base;
  int32_t dim, cap, metric;
};

// one lane's row against one query
__device__ __forceinline__ void quarantine_sums(const float *__restrict__ rp, const float *__restrict__ qp, int dim,
                                                int metric, double *o0, double *o1) {
#pragma clang fp contract(off)
  const bool cosine = metric == METRIC_COS, l2 = metric == METRIC_L2;
  double s0 = 0.0, s1 = 0.0;
  auto step = [&](float qf, float bf) {
    const double qd = (double)qf, bd = (double)bf;
    if (l2) {
      const double diff = qd - bd;
      s0 = s0 + diff * diff;
    } else {
      s0 = s0 + qd * bd;
      if (cosine) s1 = s1 + bd * bd;
    }
  };
  int i = 0;
  for (; i + 4 <= dim; i += 4) {
    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(rp + i), q4 = *reinterpret_cast<const f32x4 *>(qp + i);
    step(q4.x, b4.x);
    step(q4.y, b4.y);
    step(q4.z, b4.z);
    step(q4.w, b4.w);
  }
  for (; i < dim; ++i) step(qp[i], rp[i]);  // pad elements stay out: s + 0.0 would turn a -0.0 sum into +0.0
  *o0 = s0;
  *o1 = cosine ? s1 : 0.0;
}

static __global__ void __launch_bounds__(64) quarantine_kernel(QuarArgs a) {
  uint32_t count = a.list[0];
  if (count > (uint32_t)a.cap) count = (uint32_t)a.cap;
  const uint32_t c = blockIdx.x * 64u + threadIdx.x;
  if (c >= count) return;
  const uint32_t row = a.list[1 + c];
  BlockEntry &o = a.out[(int64_t)blockIdx.y * a.cap + c];
  o.id = a.row_base + (int64_t)row;
  quarantine_sums(a.rows + (int64_t)row * a.ld, a.Q + (int64_t)blockIdx.y * a.ldq, a.dim, a.metric, &o.s0, &o.s1);
}

// Shard mode (device candidate blocks, tsh_search_shard): the quarantined rows the caller's mask lets through
// are APPENDED to each query's block.  count keeps growing past `entries` (nothing is written there), which is
// the blocks' own overflow protocol: the merge then asks for a retry with that many entries.
struct QuarAppendArgs {
  const float *rows;
  const float *Q;
  const uint32_t *list;
  const uint64_t *mask;  // nullable: bit r = local row r may be returned
  uint8_t *blocks;       // query q's block at blocks + q * block_bytes
  int64_t ld, ldq, row_base, block_bytes;
  int32_t dim, entries, metric;
};

static __global__ void __launch_bounds__(64) quarantine_append_kernel(QuarAppendArgs a) {
  const uint32_t count = a.list[0];
  const uint32_t c = blockIdx.x * 64u + threadIdx.x;
  if (c >= count) return;
  const uint32_t row = a.list[1 + c];
  if (a.mask && !((a.mask[row >> 6] >> (row & 63)) & 1ull)) return;
  uint8_t *blk = a.blocks + (int64_t)blockIdx.y * a.block_bytes;
  const uint32_t pos = atomicAdd(&reinterpret_cast<BlockHeader *>(blk)->count, 1u);
  if (pos >= (uint32_t)a.entries) return;
  BlockEntry &o = reinterpret_cast<BlockEntry *>(blk + sizeof(BlockHeader))[pos];
  o.id = a.row_base + (int64_t)row;
  quarantine_sums(a.rows + (int64_t)row * a.ld, a.Q + (int64_t)blockIdx.y * a.ldq, a.dim, a.metric, &o.s0, &o.s1);
}

// ---------------------------------------------------------------------------
// ingest helpers
// per-row f64 norm -> inv_norm (f32), and chunk statistics for the error model
struct IngestStats {
  uint32_t max_norm_bits;  // max |row| (f32 bits, rounded up)
  uint32_t max_abs_bits;   // max finite |element|
  uint32_t nonfinite_rows; // rows holding an inf / nan element
  uint32_t tiny_rows;      // rows whose norm is nonzero but < 2^-50 (cosine error model breaks)
  uint32_t inv_min_norm_bits;  // ~(f32 bits of the smallest |row|, rounded down); 0 = no row yet
};

// irr: nullable; irr[0] = count, irr[1 ..] = local ids of IRREGULAR rows of this call -- rows the f32 error model
// cannot cover (a non-finite or > 1e15 element; for cosine a norm below 2^-50).  Up to irr_cap of them are
// listed (the host quarantines them: not live on the device, exact sums by quarantine_kernel) and stay out of the
// statistics; the rest count as before and put the shard into safe mode.
static __global__ void __launch_bounds__(256) ingest_kernel(const float *rows, int64_t ld, int dim,
                                                     int64_t first, int64_t n, float *inv_norm,
                                                     float *sqnorm, IngestStats *st, uint32_t *irr,
                                                     uint32_t irr_cap, int cosine) {
  const int lane = threadIdx.x & 63;
  int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  // per-wave running statistics, published once at the end (one atomic per
  // wave, not per row)
  uint32_t w_norm = 0, w_abs = 0, w_bad = 0, w_tiny = 0, w_inv_min = 0;
  for (int64_t r = w; r < n; r += nw) {
    const float *rp = rows + (first + r) * ld;
    double s = 0.0;
    float mx = 0.f;
    bool bad = false;
    // 16-byte loads over the padded row (pad elements are zero)
    for (int c = lane; c < (int)(ld / 4); c += 64) {
      f32x4 v4 = *reinterpret_cast<const f32x4 *>(rp + 4 * c);
      float e[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float av = fabsf(e[u]);
        if (!(av <= 3.0e38f)) bad = true;  // inf / nan
        else mx = av > mx ? av : mx;
        s += (double)e[u] * (double)e[u];
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s += __shfl_xor(s, o);
      float t = __shfl_xor(mx, o);
      mx = t > mx ? t : mx;
    }
    bad = __ballot(bad) != 0ull;
    if (lane == 0) {
      double nrm = sqrt(s);
      float inv = 0.f;
      bool tiny = false;
      if (!bad && nrm > 0.0) {
        if (nrm < 8.9e-16 /*2^-50*/) tiny = true;
        else inv = (float)(1.0 / nrm);
      }
      bool listed = false;
      if (irr && (bad || mx > 1.0e15f || (cosine && tiny))) {
        const uint32_t p = atomicAdd(&irr[0], 1u);
        if (p < irr_cap) {
          irr[1 + p] = (uint32_t)(first + r);
          listed = true;
        }
      }
      if (listed) {  // quarantined: contributes nothing on the device
        if (inv_norm) inv_norm[first + r] = 0.f;
        if (sqnorm) sqnorm[first + r] = 0.f;
        continue;
      }
      if (inv_norm) inv_norm[first + r] = inv;
      if (sqnorm) sqnorm[first + r] = (float)s;  // batched L2 key: |q|^2 + |v|^2 - 2 q.v
      float nf = (float)nrm;
      if ((double)nf < nrm) nf = __uint_as_float(__float_as_uint(nf) + 1u);
      if (!bad && __float_as_uint(nf) > w_norm) w_norm = __float_as_uint(nf);
      if (!bad) {
        const uint32_t lo = ~__float_as_uint((float)nrm > nrm ? __uint_as_float(__float_as_uint((float)nrm) - 1u) : (float)nrm);
        if (lo > w_inv_min) w_inv_min = lo;
      }
      if (__float_as_uint(mx) > w_abs) w_abs = __float_as_uint(mx);
      w_bad += bad ? 1u : 0u;
      w_tiny += tiny ? 1u : 0u;
    }
  }
  if (lane == 0) {
    if (w_norm) atomicMax(&st->max_norm_bits, w_norm);
    if (w_abs) atomicMax(&st->max_abs_bits, w_abs);
    if (w_bad) atomicAdd(&st->nonfinite_rows, w_bad);
    if (w_tiny) atomicAdd(&st->tiny_rows, w_tiny);
    if (w_inv_min) atomicMax(&st->inv_min_norm_bits, w_inv_min);
  }
}

// set / clear bits of the live bitmap (word t bit r = row t*64+r)
static __global__ void live_range_kernel(uint64_t *live, int64_t first, int64_t n, int set) {
  int64_t w0 = first >> 6, w1 = (first + n - 1) >> 6;
  for (int64_t w = w0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w <= w1;
       w += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = w * 64, hi = lo + 64;
    int64_t a = first > lo ? first : lo, b = first + n < hi ? first + n : hi;
    uint64_t m = (b - a == 64) ? ~0ull : (((1ull << (b - a)) - 1ull) << (a - lo));
    if (w == w0 || w == w1) {
      if (set) atomicOr((unsigned long long *)&live[w], (unsigned long long)m);
      else atomicAnd((unsigned long long *)&live[w], (unsigned long long)~m);
    } else {
      live[w] = set ? ~0ull : 0ull;
    }
  }
}

static __global__ void live_clear_u32_kernel(uint64_t *live, const uint32_t *ids, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    atomicAnd((unsigned long long *)&live[ids[i] >> 6], ~(1ull << (ids[i] & 63)));
}

static __global__ void live_clear_ids_kernel(uint64_t *live, const int64_t *ids, int64_t n, int64_t row_base,
                                      int64_t rows, uint32_t *n_cleared) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = ids[i] - row_base;
    if (r < 0 || r >= rows) continue;
    unsigned long long bit = 1ull << (r & 63);
    unsigned long long old = atomicAnd((unsigned long long *)&live[r >> 6], ~bit);
    if (old & bit) atomicAdd(n_cleared, 1u);
  }
}

}  // namespace tsh
