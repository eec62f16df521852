// tsh_exact.hip.h -- short searches answered in two dispatches (gfx950, wave = 64).
//
// A search that has only a few thousand rows to look at -- a selective row mask scanned as a list (config C5, keep
// 1 %: 10 k of 1 M rows), a small index (config C1: 10 k x 128), a small shard -- is not bound by HBM: the scan of a
// 1 % mask is 9-10 us, and the select (15 us, one workgroup) and the f64 re-rank (18 us) behind it are what the caller
// waits for.  The f32 keys exist to spare the f64 arithmetic on rows that cannot win; below EX_MAX_ROWS rows there is
// nothing to spare -- the exact sums of ALL of them take one launch of the re-rank's own arithmetic, spread over every
// SIMD of the device:
//   E1 exact_scan_kernel    a wave owns eight entries (rows of the list, or of the shard), eight lanes each: of every
//                           32 consecutive elements of its row a lane forms the terms of four (every term is one IEEE
//                           operation, so who forms it does not matter), and the row's sum walks through its eight
//                           lanes strictly in element order 0..d-1 -- the reference's loop
//                           (ngh_graph_engine.dart:920-946), the oracle's vs_exact_sums, rerank_kernel's arithmetic bit
//                           for bit.  Then the distance itself (L2 sqrt, IP negate, cosine 1 - dot / (|q| |v|) with
//                           similarity 0 for a non-positive denominator; f64 sqrt and divide are correctly rounded on
//                           the device as on the host) as a double.compareTo order key per entry.
//   E2 exact_select_kernel  one workgroup: the k smallest (key, position) pairs -- position = place in the ascending
//                           list = id order, the finaliser's tie-break -- by a radix select over the keys' upper
//                           halves held in registers (digits of up to eight bits from the first BIT in which they
//                           differ; once at most 64 pairs are left, one wave ranks them by whole key and position;
//                           lower halves and positions get rounds of their own only when the k-th key is tied beyond
//                           that), and their (id, s0, s1) entries + the block header where select_kernel +
//                           rerank_kernel leave theirs.  Exactly min(k, live rows) entries: no band, no overflow, no
//                           fallback.
// Everything after that (threshold, distance, sort, cut: finalize_query; the merge of shard blocks) is unchanged: it
// receives a candidate block like any other, only one without a row that cannot be a result.
#pragma once

#include "tsh_kernels.hip.h"

namespace tsh {

constexpr int EX_MAX_ROWS = 16384;  // entries E2 selects among: 16 per thread of its 1024
constexpr int EX_R = 8;             // entries per wave: eight lanes each
constexpr int EX_C = 128;           // elements of a row per piece: sixteen per lane
constexpr int EX_D = 6;             // pieces of a row in flight at once (768 elements; 96 registers)
constexpr uint64_t XKEY_DEAD = ~0ull;       // no row here (list padding, tombstone, masked out)
constexpr uint64_t XKEY_NAN = 0xFFFFFFFE00000000ull;  // a live row whose distance is NaN (double.compareTo: greatest, all
                                                      // equal): above +inf's key, its upper half below the dead marker's
constexpr uint32_t XHI_DEAD = 0xFFFFFFFFu;            // upper half of XKEY_DEAD: no live key has it

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

struct ExactArgs {
  const float *rows;     // n x ld
  const float *query;    // ld floats on the device; NULL: the query rides in the kernel-argument segment (ExactArgsQ::q)
  float *query_out;      // nullable: workgroup 0 leaves the query here (the quarantine kernels read it)
  const uint64_t *live;  // bit r of word t = row t*64+r present & not deleted & not quarantined
  const uint64_t *mask;  // nullable: caller's keep mask, same layout (entries by row; a list is the mask already)
  const uint32_t *list;  // nullable: local row ids of the entries, ascending, padded with 0xFFFFFFFF
  uint32_t *list_out;    // nullable: the list is being read from pinned host memory -- leave a device copy here (E2, later queries)
  uint64_t *xkey;        // per entry: order key of its distance (XKEY_DEAD: no row)
  double *xsum;          // per entry: s0, s1
  double sqrt_mag_a;     // cosine: sqrt of the query's sum of squares (element order, f64: query_mag_a)
  uint64_t *wmin;        // nullable: per wave (eight entries) the smallest of its keys (XKEY_DEAD: none alive), for
                         // exact_pick_kernel's bound on the k-th smallest key
  int64_t ld;            // floats per row, multiple of 4
  int64_t n_rows;        // rows of the shard
  int32_t n_entries;
  int32_t dim;
};
struct ExactArgsQ {
  ExactArgs a;
  float q[SCAN_Q_INLINE];
};
static_assert(sizeof(ExactArgsQ) <= 4096, "kernel arguments are limited to 4 KiB");

__device__ __forceinline__ uint64_t xkey_of(double d) {  // tsh_lib.hip dart_order_key, NaN one below the dead marker
  if (d != d) return XKEY_NAN;
  const uint64_t b = (uint64_t)__double_as_longlong(d);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// a 64-bit value one lane to the right / seven lanes to the left within its row of sixteen lanes (DPP: no LDS)
__device__ __forceinline__ double ex_dpp_shr1(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x111 /* row_shr:1 */, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x111, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double ex_dpp_shl7(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x107 /* row_shl:7 */, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x107, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// E1.  A wave = eight entries x eight lanes: of every 32 consecutive elements of row r, lane 8 r + g holds the four
// elements [4 g, 4 g + 4) -- a row's eight lanes load 128 contiguous bytes per instruction --, forms their terms in
// f64 (one IEEE operation each) and keeps them in registers.  A row's sum is ONE chain of dependent adds in element
// order that walks through its eight lanes: lane 0 adds its four terms, the partial sum moves one lane to the right
// (DPP), lane 1 adds its four, ... lane 7, and seven lanes back to the left for the next 32 elements.  Every lane
// executes every step (lanes that do not hold the live partial sum add their terms to a value nobody reads), so the
// instruction stream is d adds per row long whatever the lane -- the same as one lane walking the row alone, but the
// eight rows of a wave and the waves of a SIMD run side by side, and nothing goes through LDS.  All of a row's pieces
// are in flight at once (EX_D: the rows of a selective mask are a page each; a piece at a time waited out HBM +
// translation six times per row).
// Measured on the way here (16 k rows of 768 floats, one MI355X; the f32 scan of the same rows: 13 us): the terms staged
// in LDS and one chain lane per row reading them two at a time -- the CU's one LDS port was the limit -- 29 us; sixteen
// consecutive elements per lane (64-byte pieces of eight rows per load instruction: 36 cache accesses per
// instruction, SQ_WAIT_INST_ANY 10 us per wave) 26 us; a dependent v_add_f64 takes 4.1 ns, four independent ones 2.3 ns
// each, v_cvt_f64_f32 3.8 ns (tools/micro/dp_rate.hip).
template <int METRIC>
__global__ void __launch_bounds__(64) exact_scan_kernel(ExactArgsQ aq) {
#pragma clang fp contract(off)
  const ExactArgs &a = aq.a;
  const float *qsrc = a.query;
  if (!qsrc) {
    typedef const char __attribute__((address_space(4))) * karg_ptr;
    qsrc = (const float *)((karg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(ExactArgsQ, q));
  }
  const int lane = threadIdx.x;
  const int r = lane >> 3, g = lane & 7;
  const int ld = (int)a.ld, dim = a.dim;
  const int64_t e = (int64_t)blockIdx.x * EX_R + r;  // this lane's entry
  if (a.query_out && blockIdx.x == 0)
    for (int i = lane; i < ld; i += 64) a.query_out[i] = qsrc[i];
  uint32_t my_id = 0xFFFFFFFFu;
  bool alive = false;
  if (e < a.n_entries) {
    my_id = a.list ? a.list[e] : (uint32_t)e;
    if (a.list_out && g == 0) a.list_out[e] = my_id;
    if (my_id != 0xFFFFFFFFu && (int64_t)my_id < a.n_rows) {
      alive = (a.live[my_id >> 6] >> (my_id & 63)) & 1ull;
      if (alive && a.mask) alive = (a.mask[my_id >> 6] >> (my_id & 63)) & 1ull;
    }
  }
  const uint64_t am = __ballot(alive);
  const bool writer = g == 7 && e < a.n_entries;  // where a row's chain ends
  if (!am) {  // wave-uniform: nothing to read
    if (writer) a.xkey[e] = XKEY_DEAD;
    if (a.wmin && lane == 0) a.wmin[blockIdx.x] = XKEY_DEAD;
    return;
  }
  // a dead entry's lanes walk the wave's first live row (valid memory; the result is dropped)
  const uint32_t use_id = alive ? my_id : (uint32_t)__shfl((int)my_id, __builtin_ctzll(am));
  const float *rp = a.rows + (int64_t)use_id * a.ld;
  const int npiece = (dim + EX_C - 1) / EX_C;
  // rows are zero-padded to ld (a multiple of 4) and so is the query: elements in [dim, ld) are terms of +0.0, and
  // adding +0.0 changes no sum (a sum that starts at +0.0 is never -0.0).  Past ld a lane re-reads the row's first 16
  // bytes and its terms are set to +0.0.
  f32x4 raw[EX_D][4];
  auto fetch = [&](f32x4 (&dst)[4], int p) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int o = p * EX_C + 32 * c + 4 * g;  // a row's eight lanes read 128 contiguous bytes per load
      dst[c] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(rp + (o < ld ? o : 0)));
    }
  };
  auto fetch_group = [&](int p0) {
#pragma unroll
    for (int dd = 0; dd < EX_D; ++dd)
      if (p0 + dd < npiece) fetch(raw[dd], p0 + dd);  // wave-uniform
  };
  double s0 = 0.0, s1 = 0.0;
  auto piece = [&](const f32x4 (&in)[4], int p, bool group_done) {
    double t0[16], t1[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int o = p * EX_C + 32 * c + 4 * g;
      const bool ok = o < ld;
      const f32x4 qf = *reinterpret_cast<const f32x4 *>(qsrc + (ok ? o : 0));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double qd = (double)qf[i], bd = (double)in[c][i];
        double w;
        if (METRIC == METRIC_L2) {
          const double df = qd - bd;
          w = df * df;
        } else {
          w = qd * bd;
        }
        t0[4 * c + i] = ok ? w : 0.0;
        if (METRIC == METRIC_COS) t1[4 * c + i] = ok ? bd * bd : 0.0;
      }
    }
    if (group_done) fetch_group(p + 1);  // (the group's registers are free: the next group's loads, behind these terms)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 4; ++c) {  // elements 32 c + 4 step + i: four per lane, then on to the next lane
#pragma unroll
      for (int step = 0; step < 8; ++step) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          s0 = s0 + t0[4 * c + i];
          if (METRIC == METRIC_COS) s1 = s1 + t1[4 * c + i];
        }
        if (step < 7) {
          s0 = ex_dpp_shr1(s0);
          if (METRIC == METRIC_COS) s1 = ex_dpp_shr1(s1);
        }
      }
      if (c < 3 || p + 1 < npiece) {  // back to slot 0 for the next 32 elements (wave-uniform)
        s0 = ex_dpp_shl7(s0);
        if (METRIC == METRIC_COS) s1 = ex_dpp_shl7(s1);
      }
    }
  };
  fetch_group(0);
  for (int p0 = 0; p0 < npiece; p0 += EX_D) {
#pragma unroll
    for (int dd = 0; dd < EX_D; ++dd)
      if (p0 + dd < npiece) piece(raw[dd], p0 + dd, dd == EX_D - 1 && p0 + EX_D < npiece);  // wave-uniform
  }
  uint64_t key = XKEY_DEAD;
  if (writer) {
    double d;
    if (METRIC == METRIC_L2) {
      d = __builtin_sqrt(s0);
    } else if (METRIC == METRIC_IP) {
      d = -s0;
    } else {
      const double denom = a.sqrt_mag_a * __builtin_sqrt(s1);
      const double sim = denom > 0 ? s0 / denom : 0;
      d = 1.0 - sim;
    }
    key = alive ? xkey_of(d) : XKEY_DEAD;
    a.xkey[e] = key;
    *reinterpret_cast<f64x2 *>(a.xsum + 2 * e) = f64x2{s0, METRIC == METRIC_COS ? s1 : 0.0};
  }
  if (a.wmin) {  // the wave's smallest key: one plain store (a histogram of ALL keys in global memory was tried first --
    // 10 k device-scope atomics, 600 of them on the most popular bin, took this kernel from 15 to 38 us)
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
      const uint64_t other = (uint64_t)__shfl_xor((unsigned long long)key, o);
      key = other < key ? other : key;
    }
    if (lane == 7) a.wmin[blockIdx.x] = key;
  }
}

struct ExactSelArgs {
  const uint64_t *xkey;
  const double *xsum;
  const uint32_t *list;  // nullable (entry = row)
  BlockHeader *hdr;
  BlockHeader *hdr_host;  // nullable
  BlockEntry *out;
  int64_t row_base;
  int64_t shard_rows;
  int32_t n_entries;
  int32_t k;
  int32_t cap;  // >= k
  int32_t metric;
  uint32_t tag;
};
constexpr uint32_t FLAG_EXACT = 8u;  // informational: the block holds the exact top k (E1 + E2), not a band's candidates

// one more key into a 256-bin histogram in LDS: a wave whose members all name one bin (rounds among ties) adds its count
// once, otherwise lane by lane.  (A version that took up to two popular values out of every wave spent 40-60 instructions
// per key on finding them -- 3.9 us of the kernel for one round, tools/r5_x2_phases.sh; same-address queueing is not what
// a round costs: four copies of the histogram changed nothing.)
__device__ __forceinline__ void ex_hist_add(uint32_t *hist, bool in, uint32_t b, int lane) {
  const uint64_t act = __ballot(in);
  if (!act) return;  // wave-uniform (later rounds: most waves hold no member)
  const int first = __builtin_ctzll(act);
  const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)b, first);
  if (__ballot(in && b == b0) == act) {
    if (lane == first) atomicAdd(&hist[b0], (uint32_t)__popcll(act));
  } else if (in) {
    atomicAdd(&hist[b], 1u);
  }
}

constexpr int EX_FIN = 64;  // the select finishes by ranking once at most this many (key, position) pairs are left

// OR over the wave's lanes (idempotent, so every DPP step may run on all rows; no LDS): -> wave-uniform
__device__ __forceinline__ uint32_t ex_wave_or(uint32_t v) {
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, true);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141 /* row_half_mirror */, 0xF, 0xF, true);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140 /* row_mirror */, 0xF, 0xF, true);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /* row_bcast:15 */, 0xF, 0xF, true);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143 /* row_bcast:31 */, 0xF, 0xF, true);
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// E2.  The workgroup is ONE compute unit's worth of issue slots (sixteen waves on four SIMDs) and its phases are chains:
// the keys in (one global round trip), one histogram round (LDS atomics, two barriers, a scan by one wave), the ranking
// of the handful left, the entries out (a second round trip, then stores to pinned host memory).  The rounds work on the
// keys' UPPER HALVES (sign, exponent, twenty mantissa bits: one register, one-instruction compares), which decide all
// but a handful of rows; the lower halves sit in registers beside them for the ranking and for rows AT the cut.
// Measured variants, all within a microsecond of each other (14.7-18 us on average over the probe's shapes, 8-9 at
// best): whole 64-bit keys in every round (820 VALU instructions per wave); upper halves only, lower halves fetched for
// the ranking (a third round trip); this one.  What it waits for is latency, not issue.
__global__ void __launch_bounds__(1024) exact_select_kernel(ExactSelArgs a) {
  constexpr int NT = 1024, VPT = EX_MAX_ROWS / NT;
#ifndef EX_SUBH
#define EX_SUBH 4  // copies of the histogram a round's lanes spread their adds over (lane & (EX_SUBH - 1))
#endif
  __shared__ uint32_t s_hist[EX_SUBH * 256];
  __shared__ uint32_t s_bin, s_k, s_ties, s_out, s_nfin;
  __shared__ uint32_t s_part[NT / 64][3];  // per wave: live keys, OR of their upper halves, OR of the complements
  __shared__ uint32_t s_fhi[EX_FIN], s_flo[EX_FIN], s_fpos[EX_FIN];
  const int tid = threadIdx.x, lane = tid & 63;
  const int n = a.n_entries;
#ifdef TSH_PROBES  // phase stamps (100 MHz) in the header's unused fields: tools/r5_exact_try.sh
  const uint64_t pt0 = wall_clock64();
  uint64_t pt1 = 0, pt2 = 0, pa = 0, pb = 0;
  uint32_t prounds = 0, p_adds = 0, p_scan = 0;
#endif
  for (int i = tid; i < EX_SUBH * 256; i += NT) s_hist[i] = 0u;  // (clean for the first round; wave 0 clears what it reads)
  if (tid == 0) {
    s_out = 0;
    s_nfin = 0;
  }
  uint32_t h[VPT], lo[VPT];  // the keys, upper and lower halves (the rounds work on the upper ones)
  uint32_t nlive = 0;        // wave-uniform
  uint32_t o1 = 0, o0 = 0;
  // (every loop over a thread's keys stops at the entries there are: j * NT < n is workgroup-uniform -- 2000 entries
  // are two turns, not sixteen)
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    h[j] = XHI_DEAD;
    lo[j] = 0xFFFFFFFFu;
    if (j * NT >= n) continue;
    const int i = tid + j * NT;
    const uint64_t key = i < n ? a.xkey[i] : XKEY_DEAD;
    h[j] = (uint32_t)(key >> 32);
    lo[j] = (uint32_t)key;
    const bool lv = h[j] != XHI_DEAD;
    nlive += (uint32_t)__popcll(__ballot(lv));
    o1 |= lv ? h[j] : 0u;
    o0 |= lv ? ~h[j] : 0u;
  }
  {  // one barrier: every wave leaves its part, every thread reads all sixteen
    const uint32_t w1 = ex_wave_or(o1), w0 = ex_wave_or(o0);
    if (lane == 0) {
      s_part[tid >> 6][0] = nlive;
      s_part[tid >> 6][1] = w1;
      s_part[tid >> 6][2] = w0;
    }
  }
  __syncthreads();
  uint32_t live = 0, any1 = 0, any0 = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    live += s_part[w][0];
    any1 |= s_part[w][1];
    any0 |= s_part[w][2];
  }
  const uint32_t k = (uint32_t)a.k;
  const bool all = live <= k;  // no more live rows than asked for: every one of them
#ifdef TSH_PROBES
  pt1 = wall_clock64();
#endif
  // The cut: a live row is taken iff (upper half, lower half, position) <= (Khi, Klo, P) in that order.  Bits that no
  // round settled are filled with ones: every row of the last bin is taken.
  uint32_t Khi = 0xFFFFFFFFu, Klo = 0xFFFFFFFFu, P = 0xFFFFFFFFu;
  if (!all) {
    // Digits of up to eight bits from the first differing bit down (NOT byte-aligned: a corpus's distances share sign,
    // exponent and often the first mantissa bits -- aligned to bytes, the first round's keys fell into five bins and
    // their LDS atomics queued up: 5 us for that one round).  stage 0: the upper halves, 1: the lower halves of rows
    // whose upper half is the cut's, 2: the positions of rows at the cut's whole key; `up` = the stage's bits at and
    // above it are settled in its prefix.
    int stage = 0;
    const uint32_t diff = any1 & any0;  // bits in which live upper halves differ
    int up = diff ? 32 - __builtin_clz(diff) : 0;
    uint32_t pH = up >= 32 ? 0u : (any1 >> up) << up, pL = 0u, pP = 0u;  // (above `up` every live key has any1's bits)
    uint32_t kk = k, ties = live;
    auto settled = [&](uint32_t v, uint32_t pf, int u) { return u >= 32 || (v >> u) == (pf >> u); };
    auto member = [&](int j) {  // still in the running: live, and equal to what the rounds so far have settled
      const uint32_t pos = (uint32_t)(tid + j * NT);
      if (stage == 0) return h[j] != XHI_DEAD && settled(h[j], pH, up);
      if (h[j] != pH) return false;
      if (stage == 1) return settled(lo[j], pL, up);
      return lo[j] == pL && settled(pos, pP, up);
    };
    bool ranked = false;
    for (;;) {
      if (ties == kk) break;  // every row of the bin is taken
      if (ties <= (uint32_t)EX_FIN) {
        // a handful left (after the first round, usually): they are ranked by (whole key, position) -- the kk-th
        // smallest pair among them is the k-th of all.  Every wave does it for itself (sixteen times the same few
        // hundred instructions, and no second barrier to hand the result round)
#pragma unroll
        for (int j = 0; j < VPT; ++j)
          if (j * NT < n && member(j)) {
            const uint32_t slot = atomicAdd(&s_nfin, 1u);
            s_fhi[slot] = h[j];
            s_flo[slot] = lo[j];
            s_fpos[slot] = (uint32_t)(tid + j * NT);
          }
        __syncthreads();
        {
          const uint32_t m = s_nfin;  // == ties
          const uint32_t mh = (uint32_t)lane < m ? s_fhi[lane] : 0xFFFFFFFFu, ml = (uint32_t)lane < m ? s_flo[lane] : 0xFFFFFFFFu;
          const uint32_t mp = (uint32_t)lane < m ? s_fpos[lane] : 0xFFFFFFFFu;
          const unsigned long long mk = ((unsigned long long)mh << 32) | ml;
          uint32_t rank = 0;
          for (uint32_t i = 0; i < m; ++i) {  // (wave-uniform i: the other pair comes through scalar registers, not LDS)
            const unsigned long long ok = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mh, (int)i) << 32) |
                                          (uint32_t)__builtin_amdgcn_readlane((int)ml, (int)i);
            const uint32_t op = (uint32_t)__builtin_amdgcn_readlane((int)mp, (int)i);
            rank += (ok < mk) | ((ok == mk) & (op < mp));
          }
          const uint64_t hit = __ballot((uint32_t)lane < m && rank == kk - 1u);  // exactly one lane: the pairs are distinct
          const int src = __builtin_ctzll(hit);
          Khi = (uint32_t)__builtin_amdgcn_readlane((int)mh, src);
          Klo = (uint32_t)__builtin_amdgcn_readlane((int)ml, src);
          P = (uint32_t)__builtin_amdgcn_readlane((int)mp, src);
        }
        ranked = true;
        break;
      }
      if (up == 0) {  // this stage is settled and its ties go beyond k: on to the next one
        if (stage == 0) {
          up = 32;
        } else {
          up = 14;  // positions are below EX_MAX_ROWS = 2^14; distinct, so stage 2 ends with ties == kk == 1
        }
        ++stage;
      }
#ifdef TSH_PROBES
      ++prounds;
      pa = wall_clock64();
#endif
      const int lob = up > 8 ? up - 8 : 0;  // this round's digit: bits [lob, up)
      const uint32_t dmask = (1u << (up - lob)) - 1u;
#pragma unroll
      for (int j = 0; j < VPT; ++j) {
        if (j * NT >= n) continue;
        const uint32_t v = stage == 0 ? h[j] : (stage == 1 ? lo[j] : (uint32_t)(tid + j * NT));
        ex_hist_add(s_hist + 256 * (lane & (EX_SUBH - 1)), member(j), (v >> lob) & dmask, lane);
      }
      __syncthreads();
#ifdef TSH_PROBES
      pb = wall_clock64();
      p_adds += (uint32_t)(pb - pa);
#endif
      if (tid < 64) {  // wave 0: lane l owns bins 4l .. 4l+3 (block_kth_radix's scan)
        uint32_t c[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          c[j] = 0u;
#pragma unroll
          for (int sub = 0; sub < EX_SUBH; ++sub) {
            c[j] += s_hist[256 * sub + 4 * lane + j];
            s_hist[256 * sub + 4 * lane + j] = 0u;
          }
        }
        const uint32_t mine = c[0] + c[1] + c[2] + c[3];
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t u = (uint32_t)__shfl_up((int)incl, d);
          if (lane >= d) incl += u;
        }
        uint32_t below = incl - mine;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (kk > below && kk <= below + c[j]) {  // exactly one (lane, j): 1 <= kk <= number of keys in the round
            s_bin = (uint32_t)(4 * lane + j);
            s_k = kk - below;
            s_ties = c[j];
          }
          below += c[j];
        }
      }
      __syncthreads();
#ifdef TSH_PROBES
      p_scan += (uint32_t)(wall_clock64() - pb);
#endif
      const uint32_t digit = s_bin << lob;
      if (stage == 0) pH |= digit;
      else if (stage == 1) pL |= digit;
      else pP |= digit;
      up = lob;
      kk = s_k;
      ties = s_ties;
    }
    if (!ranked) {  // the cut lies behind the last bin taken: unsettled bits of its stage, and the later stages, all ones
      const uint32_t fill = up >= 32 ? 0xFFFFFFFFu : (1u << up) - 1u;
      Khi = stage == 0 ? pH | fill : pH;
      Klo = stage == 0 ? 0xFFFFFFFFu : (stage == 1 ? pL | fill : pL);
      P = stage == 2 ? pP | fill : 0xFFFFFFFFu;
    }
  }
#ifdef TSH_PROBES
  pt2 = wall_clock64();
#endif
  // the entries: (id, s0, s1) of every taken row, in no particular order (the finaliser sorts)
  const bool need_lo = !(Klo == 0xFFFFFFFFu && P == 0xFFFFFFFFu);  // workgroup-uniform
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    if (j * NT >= n) continue;
    const uint32_t pos = (uint32_t)(tid + j * NT);
    const bool lv = h[j] != XHI_DEAD;
    bool take = lv && h[j] <= Khi;
    if (need_lo && lv && h[j] == Khi) take = lo[j] < Klo || (lo[j] == Klo && pos <= P);  // rows AT the cut's upper half
    const uint64_t bm = __ballot(take);
    if (bm) {  // wave-uniform
      uint32_t base = 0;
      if (lane == __builtin_ctzll(bm)) base = atomicAdd(&s_out, (uint32_t)__popcll(bm));
      base = (uint32_t)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(bm));
      if (take) {
        const uint32_t p = base + (uint32_t)__popcll(bm & ((1ull << lane) - 1ull));
        if (p < (uint32_t)a.cap) {
          const uint32_t row = a.list ? a.list[pos] : pos;
          const f64x2 sv = *reinterpret_cast<const f64x2 *>(a.xsum + 2 * (int64_t)pos);
          BlockEntry e;
          e.id = a.row_base + (int64_t)row;
          e.s0 = sv.x;
          e.s1 = sv.y;
          a.out[p] = e;
        }
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    BlockHeader hv;
    hv.count = s_out;
    hv.entries = (uint32_t)a.cap;
    hv.tau_key = KEY_NAN;   // (no f32 keys exist for this block)
    hv.band_key = KEY_NAN;
    hv.tiles_hit = 0u;
#ifdef TSH_PROBES
    hv.tau_key = (uint32_t)(pt1 - pt0);               // keys in, live count, range
    hv.band_key = (uint32_t)(pt2 - pt1);              // the select
    hv.tiles_hit = (uint32_t)(wall_clock64() - pt2);  // the entries
#endif
    hv.flags = FLAG_EXACT | (s_out > (uint32_t)a.cap ? FLAG_LIST_OVERFLOW : 0u);  // (cap >= k: never)
    hv.k = (uint32_t)a.k;
    hv.metric = (uint32_t)a.metric;
    hv.row_base = a.row_base;
    hv.shard_rows = a.shard_rows;
    hv.pad[0] = hv.pad[2] = hv.pad[3] = 0u;
#ifdef TSH_PROBES
    hv.pad[2] = prounds;
    hv.pad[0] = (p_adds << 16) | (p_scan & 0xFFFFu);  // (ticks of 10 ns: the rounds' adds, their scans)
    hv.pad[3] = s_nfin;
#endif
    hv.pad[1] = a.tag;
    *a.hdr = hv;
    if (a.hdr_host) *a.hdr_host = hv;
  }
}

// E2' (round 6).  exact_select_kernel above is ONE workgroup, and what it costs grows with the entries: every phase of it
// is a loop over a thread's key slots -- 8.6 us at 2 k entries, 12.8-17 at 10 k (profiles/r06_lone_trace.txt), on one
// compute unit of 256, behind a scan that took 6-15.  The wide pick spreads the per-entry work over one workgroup per
// 256 entries and cuts the cross-workgroup dependency instead of synchronising over it:
//   * a BOUND instead of the k-th key: E1 leaves every wave's smallest key (eight entries; one plain store).  The k-th
//     smallest of those minima is at or above the k-th smallest key (k waves hold a key at or below it), and hardly
//     above it: with G waves about 0.44 k^2 / G rows lie between the two (3.5 at k = 100 of 10 k rows).  Every workgroup
//     finds it for itself -- up to 2048 minima, a histogram in its own LDS over eleven bits from the first bit in which
//     the minima's upper halves differ, one scan -- and takes the cut bin's upper edge;
//   * no ranking: every row at or below the bound is emitted -- k, the rows up to the k-th minimum, and what shares the
//     cut bin (1/2048 of the minima's range).  The block then holds a few rows that cannot win, like a pre-filter block
//     does, and the finaliser's threshold / sort / cut (ngh_graph_engine.dart:127,133-134) is what it always was;
//   * ONE device-scope atomic per workgroup: a 64-bit add whose low half hands out output positions and whose high half
//     is a ticket -- the workgroup that draws the last ticket knows the final count and writes the header.  (Per-wave
//     position atomics, 80 of them on one address, took a 40-workgroup launch from 4.3 to 8.8 us.)
//   * a bound that lets in more rows than the block holds (ties by the hundred; fewer live waves than k) sets
//     FLAG_LIST_OVERFLOW, and the host finishes the search with exact_select_kernel on the same keys (job_finish).
constexpr int EX_PICK_BINS = 2048, EX_PICK_MPT = EX_MAX_ROWS / EX_R / 256;  // minima per thread: 8
struct ExactPickArgs {
  const uint64_t *xkey;
  const double *xsum;
  const uint32_t *list;  // nullable (entry = row)
  const uint64_t *wmin;  // E1's wave minima: n_groups of them
  unsigned long long *ctr;  // zero between searches: low half = entries handed out, high half = workgroups done
  BlockHeader *hdr;
  BlockHeader *hdr_host;  // nullable
  BlockEntry *out;
  int64_t row_base;
  int64_t shard_rows;
  int32_t n_entries;
  int32_t n_groups;
  int32_t k;
  int32_t cap;
  int32_t metric;
  uint32_t tag;
};

__global__ void __launch_bounds__(256) exact_pick_kernel(ExactPickArgs a) {
  __shared__ uint32_t s_hist[EX_PICK_BINS];
  __shared__ uint32_t s_part[4][3], s_tot[4], s_take[4], s_cut, s_ck, s_kk, s_base, s_final;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = blockIdx.x * 256 + tid;
  // this thread's entry, all of it, before the bound is known (one round trip instead of two)
  const uint64_t key = i < a.n_entries ? a.xkey[i] : XKEY_DEAD;
  const uint32_t h = (uint32_t)(key >> 32);
  const bool live = h != XHI_DEAD;
  uint32_t row = (uint32_t)i;
  f64x2 sv = f64x2{0.0, 0.0};
  if (live) {
    if (a.list) row = a.list[i];
    sv = *reinterpret_cast<const f64x2 *>(a.xsum + 2 * (int64_t)i);
  }
  // the minima's upper halves; which bits they differ in; how many are alive
  uint32_t m[EX_PICK_MPT];
  uint32_t o1 = 0, o0 = 0, nl = 0;
#pragma unroll
  for (int j = 0; j < EX_PICK_MPT; ++j) {
    const int g = tid + 256 * j;
    m[j] = XHI_DEAD;
    if (j * 256 < a.n_groups && g < a.n_groups) m[j] = (uint32_t)(a.wmin[g] >> 32);
    const bool lv = m[j] != XHI_DEAD;
    nl += (uint32_t)__popcll(__ballot(lv));
    o1 |= lv ? m[j] : 0u;
    o0 |= lv ? ~m[j] : 0u;
    s_hist[tid + 256 * j] = 0u;  // (EX_PICK_BINS = 256 * EX_PICK_MPT)
  }
  {
    const uint32_t w1 = ex_wave_or(o1), w0 = ex_wave_or(o0);
    if (lane == 0) {
      s_part[wave][0] = nl;
      s_part[wave][1] = w1;
      s_part[wave][2] = w0;
    }
  }
  __syncthreads();
  uint32_t groups = 0, any1 = 0, any0 = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    groups += s_part[w][0];
    any1 |= s_part[w][1];
    any0 |= s_part[w][2];
  }
  const uint32_t k = (uint32_t)a.k;
  uint32_t Khi = 0xFFFFFFFEu;  // fewer live waves than k: no bound -- every live row
  if (groups >= k) {           // (workgroup-uniform)
    // eleven bits a round from the first bit in which the minima differ; a cut bin that still holds more than a handful
    // of minima is refined (a corpus whose distances change sign -- inner products -- differs in the sign bit: the first
    // round's bins are then two octaves wide, and the k-th minimum shares its bin with hundreds)
    const uint32_t diff = any1 & any0;
    int up = diff ? 32 - __builtin_clz(diff) : 0, lob = 0;
    uint32_t prefix = up >= 32 ? 0u : (any1 >> up) << up, kk = k;
    for (int round = 0;; ++round) {
      lob = up > 11 ? up - 11 : 0;
      const uint32_t dmask = (1u << (up - lob)) - 1u;
      if (round) {
#pragma unroll
        for (int j = 0; j < EX_PICK_MPT; ++j) s_hist[tid + 256 * j] = 0u;
        __syncthreads();
      }
#pragma unroll
      for (int j = 0; j < EX_PICK_MPT; ++j)
        if (m[j] != XHI_DEAD && (up >= 32 || (m[j] >> up) == (prefix >> up))) atomicAdd(&s_hist[(m[j] >> lob) & dmask], 1u);
      __syncthreads();
      uint32_t c[8], mine = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        c[j] = s_hist[8 * tid + j];
        mine += c[j];
      }
      uint32_t incl = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += u;
      }
      if (lane == 63) s_tot[wave] = incl;
      __syncthreads();
      uint32_t below = incl - mine;
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if (w < wave) below += s_tot[w];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (below < kk && below + c[j] >= kk) {  // exactly one (thread, j): 1 <= kk <= members of the round
          s_cut = (uint32_t)(8 * tid + j);
          s_ck = c[j];
          s_kk = kk - below;
        }
        below += c[j];
      }
      __syncthreads();
      prefix |= s_cut << lob;
      if (lob == 0 || s_ck <= 8u) break;  // (workgroup-uniform)
      kk = s_kk;
      up = lob;
    }
    Khi = prefix | ((1u << lob) - 1u);  // the cut bin's upper edge
  }
  const bool take = live && h <= Khi;
  const uint64_t bm = __ballot(take);
  if (lane == 0) s_take[wave] = (uint32_t)__popcll(bm);
  __syncthreads();
  uint32_t total = 0, before = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < wave) before += s_take[w];
    total += s_take[w];
  }
  if (tid == 0) {
    const unsigned long long old = atomicAdd(a.ctr, (1ull << 32) | (unsigned long long)total);
    s_base = (uint32_t)old;
    s_final = (uint32_t)(old >> 32) == gridDim.x - 1u ? (uint32_t)old + total : 0xFFFFFFFFu;
    if ((uint32_t)(old >> 32) == gridDim.x - 1u) *a.ctr = 0ull;  // (nobody adds after the last ticket: clear for the next search)
  }
  __syncthreads();
  if (take) {
    const uint32_t p = s_base + before + (uint32_t)__popcll(bm & ((1ull << lane) - 1ull));
    if (p < (uint32_t)a.cap) {
      BlockEntry e;
      e.id = a.row_base + (int64_t)row;
      e.s0 = sv.x;
      e.s1 = sv.y;
      a.out[p] = e;
    }
  }
  if (tid == 0 && s_final != 0xFFFFFFFFu) {  // the last workgroup: every other one has booked its rows
    BlockHeader hv;
    hv.count = s_final;
    hv.entries = (uint32_t)a.cap;
    hv.tau_key = KEY_NAN;  // (no f32 keys exist for this block)
    hv.band_key = KEY_NAN;
    hv.tiles_hit = 0u;
    hv.flags = FLAG_EXACT | (s_final > (uint32_t)a.cap ? FLAG_LIST_OVERFLOW : 0u);
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

}  // namespace tsh
