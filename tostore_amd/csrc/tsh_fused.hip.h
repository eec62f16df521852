// tsh_fused.hip.h -- ONE dispatch per query for short scans: K1 scan, K2 select and K4 re-rank of
// tsh_kernels.hip.h in a single kernel.
//
// A query over a small shard (config C1: 10 k x 128; the row range one GPU of an 8-GPU index holds) or behind a
// selective row mask (config C5, keep 1 %) scans for a few microseconds; three dependent launches, the gaps
// between them and the host's share of each were most of its latency (C1: 20 us of kernels in a 37 us call).
// Here every workgroup scans its tiles, then takes a ticket; the workgroup that draws the last ticket has all
// keys / tile minima in memory (release by every producer, acquire by the one consumer: MI355X_MICROARCH.md,
// "Workgroup dispatch, XCD placement & inter-workgroup visibility"), selects the candidates and re-ranks them
// exactly -- no grid barrier, no co-residency requirement, nothing spins.
//
// The scan here is the GENERIC form of scan_kernel: row width is a run-time value, the query sits in LDS instead
// of registers, eight rows are in flight per wave (one 16-byte load each per 256-float column block).  It gives
// up the last 15-20 % of scan_kernel's HBM rate, which is why the host only takes this path when the scan is
// short (FUSED_MAX_SCAN_BYTES in tsh_lib.hip) and latency, not bandwidth, is what the caller waits for.
// Same keys (per-lane partial sums over the same column blocks, same transposing butterfly), same select, same
// re-rank arithmetic: results are bit-identical to the three-kernel path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tsh_batch.hip.h"
#include "tsh_kernels.hip.h"

namespace tsh {

struct FusedArgs {
  ScanArgs scan;       // query == nullptr: the query rides in the kernel arguments (FusedArgsQ::q)
  SelectArgs sel;
  RerankArgs rr;
  uint32_t *ticket;    // device, zero on entry; the last workgroup leaves it zero again
};
constexpr int FUSED_Q_INLINE = 768;  // query floats in the kernel-argument segment (larger queries: scan.query)
struct FusedArgsQ {
  FusedArgs a;
  float q[FUSED_Q_INLINE];
};
static_assert(sizeof(FusedArgsQ) <= 4096, "kernel arguments are limited to 4 KiB");

constexpr int FUSED_MAX_D4 = 1024;  // dim <= 4096

// NT = threads per workgroup: 256, or 1024 when the select has more than 4096 tile minima to look at.
template <int NT, int METRIC, bool MASKED>
__global__ void __launch_bounds__(NT) fused_query_kernel(FusedArgsQ aq) {
  const ScanArgs &a = aq.a.scan;
  __shared__ __attribute__((aligned(16))) f32x4 s_q[FUSED_MAX_D4];
  __shared__ uint32_t s_last;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int WPB = NT / 64;
  const int d4 = a.d4;
  {
    const float *qsrc = a.query;
    if (!qsrc) {
      typedef const char __attribute__((address_space(4))) * karg_ptr;
      qsrc = (const float *)((karg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(FusedArgsQ, q));
    }
    for (int c = tid; c < d4; c += NT) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(qsrc + 4 * c);
      s_q[c] = v;
      if (a.query_out && blockIdx.x == 0) *reinterpret_cast<f32x4 *>(a.query_out + 4 * c) = v;
    }
  }
  __syncthreads();

  // ---- K1, generic: a wave owns a tile of 64 rows; lane l holds 16 bytes of every 256-float column block ---------
  const int nblk = (d4 + 63) >> 6;  // column blocks per row (wave-uniform)
  const int stride = (int)gridDim.x * WPB;
  for (int t = MASKED ? wave * (int)gridDim.x + (int)blockIdx.x : (int)blockIdx.x * WPB + wave; t < a.n_tiles; t += stride) {
    const float *tbase = a.rows + (int64_t)t * 64 * a.ld;
    uint64_t bits = ~0ull;
    int cnt = 64;
    if (MASKED) {
      uint64_t w = a.live[t];
      if (a.mask) w &= a.mask[t];
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)w);
      const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(w >> 32));
      bits = ((uint64_t)hi << 32) | lo;
      cnt = __popcll(bits);
      if (cnt == 0) {
        if (lane == 0) a.gmin[t] = KEY_DEAD;  // keys[] of a dead tile stay stale: every reader checks gmin first
        continue;
      }
    }
    const int nb = MASKED ? (cnt + 7) >> 3 : 8;  // 8-row batches, wave-uniform
    uint64_t rem = bits;
    int last = 0;
    float val = 0.f;
    for (int b = 0; b < nb; ++b) {
      int rowi[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (MASKED) {  // next live row (a short last batch repeats the final row)
          if (rem) {
            last = __builtin_ctzll(rem);
            rem &= rem - 1;
          }
          rowi[j] = last;
        } else {
          rowi[j] = b * 8 + j;
        }
      }
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
      for (int cb = 0; cb < nblk; ++cb) {
        const int c = cb * 64 + lane;
        const bool has = c < d4;  // lanes past the row end re-read the row's first 16 bytes (valid memory), times 0
        const f32x4 q4 = has ? s_q[c] : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          v[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(tbase + (int64_t)rowi[j] * a.ld + (has ? 4 * c : 0)));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          f32x4 x = v[j];
          if (!has) x = f32x4{0.f, 0.f, 0.f, 0.f};
          acc[j] = accum4<METRIC>(acc[j], q4, x);
        }
      }
      float o = treduce8<0>(acc, lane);
      o += __shfl_xor(o, 8);
      o += __shfl_xor(o, 16);
      o += __shfl_xor(o, 32);
      if ((lane >> 3) == b) val = o;  // slot b*8 + (lane&7) == lane
    }
    bool alive;
    if (MASKED) {  // expand compact slots back to row positions
      const uint64_t below = bits & ((1ull << lane) - 1ull);
      val = __shfl(val, __popcll(below));
      alive = (bits >> lane) & 1ull;
    } else {
      alive = (int64_t)t * 64 + lane < a.n;
    }
    if (METRIC == METRIC_IP) {
      val = -val;
    } else if (METRIC == METRIC_COS) {
      const float inv = alive ? a.inv_norm[(int64_t)t * 64 + lane] : 0.f;
      val = -(val * inv);
    }
    const uint32_t key = alive ? f2key(val) : KEY_DEAD;
    a.keys[(int64_t)t * 64 + lane] = key;
    const uint32_t m = wave_min_u32(key);
    if (lane == 0) a.gmin[t] = m;
  }

  // ---- ticket: the last workgroup to get here carries on --------------------------------------------------------
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // this workgroup's keys / gmin (and query_out) are visible
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the compiler may drop the wait behind the write-back)
    const uint32_t t = __hip_atomic_fetch_add(aq.a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last_one = t + 1u == gridDim.x;
    if (last_one) {
      __hip_atomic_store(aq.a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next query
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop this CU's stale L1 lines
    }
    s_last = last_one ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;

  // ---- K2 + K4 by the last workgroup -----------------------------------------------------------------------------
  select_body<NT, true>(aq.a.sel);
  __syncthreads();  // header and candidate rows (global memory, written by this workgroup) are ordered for it
  const RerankArgs &r = aq.a.rr;
  uint32_t count = *r.count_ptr;
  if (count > (uint32_t)r.cap) count = (uint32_t)r.cap;
  for (uint32_t c0 = (uint32_t)wave * 64u; c0 < count; c0 += NT) {  // wave-uniform: 64 candidates per wave and round
    const bool mine = c0 + lane < count;
    const uint32_t row = r.cand_rows[mine ? c0 + lane : c0];
    double s0, s1;
    rerank_lane_sums(r.rows + (int64_t)row * r.ld, r.query, r.dim, (int)r.ld, r.metric, &s0, &s1);
    if (mine) {
      r.out[c0 + lane].id = r.row_base + (int64_t)row;
      r.out[c0 + lane].s0 = s0;
      r.out[c0 + lane].s1 = s1;
    }
  }
}

}  // namespace tsh
