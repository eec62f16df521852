// tsh_batch_f16pp.hip.h -- the fp16 pre-filter key kernel of the batched path, third generation ("ping-pong").
// Same contract, operand planes ("plane32") and list protocol as batch_score_f16_kernel (tsh_batch_f16.hip.h);
// what changed is who does what when:
//
//   * the two waves that share a SIMD never multiply at the same time.  Waves 0-3 (group 0: the upper half of the
//     tile's queries) and waves 4-7 (group 1: the lower half; wave w + 4 sits on wave w's SIMD) run the same
//     program one s_barrier apart:   LOAD(c) | barrier | COMPUTE(c) | barrier | LOAD(c + 1) ...
//     so that group 0 multiplies chunk c while group 1 reads chunk c's fragments and issues its share of the DMA,
//     and the other way round in the next phase.  A COMPUTE phase is nothing but 4 MI back-to-back MFMAs (512 matrix
//     cycles at MI = 4); a LOAD phase is the 3 MI ds_read_b128 of the wave's fragments, STAGE / 8 KB LDS-DMA pieces
//     and the waits -- everything a wave does besides multiplying happens while its SIMD partner owns the matrix
//     pipe, and the barrier's latency hides behind the last MFMA of the phase.  (Second generation: both waves
//     interleave reads, DMA issue and MFMAs, and all eight meet at one barrier per K-step: 1550-1800 cycles per step
//     for 1024 cycles of MFMA issue.)
//   * fragments are single-buffered (48 registers instead of 96): read in LOAD, consumed in COMPUTE.
//   * every wave issues its share of the stream (4 pieces per chunk at MI = 4), three chunks ahead, and waits for
//     its own pieces of chunk c + 1 with a counted vmcnt at the END of LOAD(c) -- two whole iterations after their
//     issue.
//   * In the filtered pass the accumulators START at -theta_q (theta_q = the query's threshold in accumulator units),
//     so "this row passes" is the accumulator's sign: a 32 x 32 block is tested with eight v_max3_i32 and one
//     compare, and only blocks with a survivor (3.7 of a wave's eight per tile at k = 100) are looked at register by
//     register (epilogue, below).  The key stored for a survivor is (seed - acc) * scale; the error model's
//     accumulation term doubles (the chain now carries |theta| <= |q| max|v| as well: tsh_host_batch.inl.h,
//     batch_delta2).  Thresholds are capped at the largest key a row can have (BatchArgs::kmax), so an "everything
//     passes" threshold stays finite.  The dense (sample) pass starts at zero.
//   * L2 (round 4): key = |q|^2 + |v|^2 - 2 q.v <= thr  <=>  dot_acc - |v|^2 / (2 s) - (|q|^2 - thr) / (2 s) >= 0
//     (s = the power-of-two dot scale).  The per-query term is the seed as before; the per-ROW term is one value per
//     lane and column block in the 32 x 32 MFMA's C layout (column = lane & 31 = corpus row), so the accumulators
//     start at seed_q - c_v with c_v = |v|^2 / (2 s): a v_sub where IP / cosine have a v_mov, and everything after
//     it -- sign test, notes, appends -- is the same code.  A survivor's key is thr'_q - 2 s acc (thr' = the capped
//     threshold).  c_v of the NEXT tile's rows is loaded in the epilogue, beside the liveness words (a compiler-
//     tracked global load in the K loop would drain the DMA ring with its vmcnt wait).
//   * thresholds live in per-wave LDS tables (a wave only needs its own 32 MI query rows): nothing in the kernel
//     needs a workgroup-wide barrier besides the phase barriers, whose count is the same for every wave.  After a
//     tile's last chunk group 0 passes the phase barrier before its epilogue, group 1 after its own: the two
//     epilogues of a SIMD run side by side.
//
// What bounds it (probe builds, tools/r3_pp_variants.sh; 1 M x 768, 1024 queries, key passes without epilogue):
// no loads at all 0.89 ms (the matrix pipes alone: 0.85); the same 32 DMA instructions per K-step moving 4 B per lane
// instead of 16: 1.08; 16 B per lane into registers instead of LDS: 1.08; as shipped: 1.33.  Where the pieces are
// issued (LOAD phase, between the MFMAs, any split), whether their data is L2- or L1-hot, and whether anybody waits
// for them makes no difference: a K-step pays ~8 cycles per DMA instruction and ~14 cycles per KB written into LDS,
// on top of the matrix time -- the LDS side of this tile shape, not the memory side.
// Tried on top of that and dropped: the corpus rows' fragments straight from the plane into registers (buffer loads,
// three chunks ahead, four register sets; only the queries through the DMA ring -- half the LDS writes).  With
// nobody waiting for those loads the K loop runs in 1.19 ms (PP_ISSUE=4); with the MFMAs actually consuming them
// 1.38-1.44 (1.31 with a plane layout that makes each load a contiguous 1 KB, 1.27 with the rows L2-hot) against
// 1.32-1.34 for the ring: a load that misses L2 now stalls a COMPUTE phase directly, and 256 registers leave no room
// for a deeper prefetch.  768-byte DMA pieces (global_load_lds_dwordx3): no faster.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tsh_batch_f16.hip.h"

namespace tsh {

#ifdef TSH_PROBES
// probes: the same instruction stream with 4 B per lane into LDS / with 16 B per lane into registers
__device__ __forceinline__ void f16_dma4_s(const void *sbase_any, uint32_t voff, uint32_t lds_dst) {
  const uint64_t pb = (uint64_t)sbase_any;
  const void *sbase = reinterpret_cast<const void *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pb >> 32)) << 32) |
                                                     (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pb));
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ void f16_load16_s(const void *sbase_any, uint32_t voff, u32x4 *sink) {
  const uint64_t pb = (uint64_t)sbase_any;
  const void *sbase = reinterpret_cast<const void *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pb >> 32)) << 32) |
                                                     (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pb));
  asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(*sink) : "v"(voff), "s"(sbase) : "memory");
}
#endif

template <int METRIC, bool DENSE, int MI>
__global__ void __launch_bounds__(512, 2) batch_score_f16pp_kernel(BatchArgs a) {
  static_assert(MI == 4 || MI == 2, "wave patch 32 MI x 64; 8 waves as 2 x 4");
  constexpr int WN = 4, TM = 64 * MI, TN = 256;
  constexpr int STAGE = (TM + TN) * 64;            // bytes: [queries TM x 64 B][rows TN x 64 B] = 32 / 24 KB
  constexpr int NST = 4;                           // ring stages
  constexpr int PPW = STAGE / 1024 / 8;            // 1 KB DMA pieces per wave and chunk: 4 / 3
  constexpr int NPC = PPW;                         // DMA instructions per wave and chunk
#ifdef PP_K
  constexpr int PK = PP_K < NPC ? PP_K : NPC;      // ... of which a wave issues PK in its LOAD phase, the others in COMPUTE
#else
  constexpr int PK = NPC;
#endif
  auto piece_off = [](int u) -> int { return u * 1024; };
  constexpr int QROWS = 32 * MI;                   // query rows of a wave's patch
  constexpr bool L2 = METRIC == METRIC_L2;  // (cosine planes hold unit rows: its key is -dot, as for IP)
  constexpr bool SEEDED = !DENSE;
  // Round 5: the fp16 error band of an L2 / inner-product key is PER ROW -- alpha_q |v| + beta_q (batch_delta2) -- and
  // the filtered pass tests and stores the key's LOWER side, key - alpha_q |v|: one more per-(query, row) term in the
  // accumulators' start values, a v_fma where L2 had a v_sub and inner product a v_mov.
  constexpr bool ROWW = SEEDED && METRIC != METRIC_COS;
  // ... and the dense (sample) pass writes the key's UPPER side, key + alpha_q |v| (what the threshold is an order
  // statistic of), when the host hands it alpha (the probes of the error model read the plain keys: no alpha there)
  constexpr bool ROWV = METRIC != METRIC_COS;  // the rows' norms are wanted (either pass)
  __shared__ __attribute__((aligned(1024))) unsigned char ring[NST * STAGE];
  __shared__ __attribute__((aligned(16))) float s_seed[8][QROWS];  // per wave: -theta of its query rows
  __shared__ __attribute__((aligned(16))) float s_aux[8][L2 ? QROWS : 4];  // L2: thr' (filtered) resp. |q|^2 (dense)
  __shared__ __attribute__((aligned(16))) float s_alpha[8][ROWV ? QROWS : 4];  // per wave: alpha_q (filtered pass: in accumulator units)
  __shared__ uint2 s_hits[8][F16_HITS + 64];  // + one spare slot per lane

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;  // wm = ping-pong group
  const int KC = a.hchunks;
  const int G = (int)gridDim.x, total_tiles = a.q_tiles * a.n_tiles;

  // ---- issue stream ------------------------------------------------------------------------------------------
  // The stage image is STAGE / 1 KB pieces, piece pi = block pi / 8 (blocks 0 .. TM / 128 - 1: the tile's query
  // groups, then its two row groups), 1 KB pi % 8 of that block; wave w moves pieces w PPW .. w PPW + PPW - 1 of
  // every chunk.  A block is contiguous in the planes, chunk after chunk 8 KB apart.
  const unsigned char *src[NPC];  // wave-uniform: my pieces of the next chunk of the stream
  int i_tile = (int)blockIdx.x;
  auto set_src = [&](int tile) {
    int qt, nt;
    batch_tile_of(a, tile, &qt, &nt);
#ifdef TSH_PROBES
    if (a.dbg & 8) nt &= 7;  // probe: every workgroup streams the same few row tiles (L2-hot; results are wrong)
#endif
    const int qb = qt * TM, nb = a.row0 + nt * TN;  // row0 is a multiple of the tile (host)
#pragma unroll
    for (int u = 0; u < NPC; ++u) {
      const int pi = wave * PPW + u, blk = pi >> 3, pc = pi & 7;
      const bool is_q = blk < TM / F16_GROUP;
      const int64_t group = is_q ? (qb / F16_GROUP + blk) : (nb / F16_GROUP + (blk - TM / F16_GROUP));
      const uint64_t p = (uint64_t)(reinterpret_cast<const unsigned char *>(is_q ? a.Qs : a.Vs) + (group * KC) * 8192 + pc * 1024);
      src[u] = reinterpret_cast<const unsigned char *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(p >> 32)) << 32) |
                                                       (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)p));
    }
  };
#ifdef TSH_PROBES
  u32x4 dbg_sink = {0u, 0u, 0u, 0u};
#endif
  const uint32_t lane_off = (uint32_t)lane * 16u;
  const uint32_t my_dst = (uint32_t)(uintptr_t)ring + (uint32_t)(wave * PPW) * 1024u;
  // (one piece: the K loop spreads a chunk's pieces over the LOAD and the COMPUTE phase)
  auto issue_one = [&](auto SIDX, int u) {
    constexpr int ST = decltype(SIDX)::value;
#if defined(TSH_PROBES) && defined(PP_ISSUE) && PP_ISSUE == 1    // probe: 4 B per lane into LDS
    f16_dma4_s(src[u], lane_off, my_dst + (uint32_t)(ST * STAGE + piece_off(u)));
#elif defined(TSH_PROBES) && defined(PP_ISSUE) && PP_ISSUE == 2  // probe: 16 B per lane into registers
    f16_load16_s(src[u], lane_off, &dbg_sink);
#elif defined(TSH_PROBES) && defined(PP_ISSUE) && PP_ISSUE == 3  // probe: no loads
#elif defined(TSH_PROBES) && defined(PP_ISSUE) && PP_ISSUE == 4  // probe: what "rows straight into registers" would cost
    // at best: the query half of the stage by DMA as shipped, the row half not at all; instead EVERY wave loads 4 KB
    // (its own 64 rows' chunk) into registers, 16 B per lane, and nobody waits for them (results are wrong)
    if (wave * PPW + u < (TM / F16_GROUP) * 8) f16_dma16_s(src[u], lane_off, my_dst + (uint32_t)(ST * STAGE + piece_off(u)));
    f16_load16_s(src[u], lane_off, &dbg_sink);
#else
    f16_dma16_s(src[u], lane_off, my_dst + (uint32_t)(ST * STAGE + piece_off(u)));
#endif
    src[u] += 8192;
  };
  auto issue = [&](auto SIDX) {  // a whole chunk (prologue)
#pragma unroll
    for (int u = 0; u < NPC; ++u) issue_one(SIDX, u);
  };
  auto next_tile_src = [&]() {  // the stream moves on to my next tile (or stays on the last one: surplus re-reads
    if (i_tile + G < total_tiles) i_tile += G;  // into stages nobody reads any more keep the vmcnt counts constant)
    set_src(i_tile);
  };

  // fragment addresses: row = 32 i + (lane & 31) (+ patch base), logical piece 2 s + (lane >> 5), slot = piece ^ sw
  const int half = lane >> 5, l31 = lane & 31, sw = (l31 >> 2) & 3;
  const int a_off = (wm * QROWS + l31) * 64, b_off = TM * 64 + (wn * 64 + l31) * 64;
  const int slot0 = ((0 + half) ^ sw) * 16, slot1 = ((2 + half) ^ sw) * 16;
  f32x16 acc[MI][2];
  f16x8 fa[2][MI], fb[2][2];  // the two k16 slabs of ONE chunk: read in LOAD, multiplied in COMPUTE
  auto read_frags = [&](auto SIDX) {
    constexpr int ST = decltype(SIDX)::value;
    const unsigned char *st = ring + ST * STAGE;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      const int so = sl == 0 ? slot0 : slot1;
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[sl][i] = *reinterpret_cast<const f16x8 *>(st + a_off + i * 2048 + so);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[sl][j] = *reinterpret_cast<const f16x8 *>(st + b_off + j * 2048 + so);
    }
  };

  // probe (TSH_PROBES builds, TSH_F16_DBG & 32): shader-clock stamps of waves 0, 4, 1, 5 of workgroup 0 over their
  // first 96 chunks; word 11 of a wave's first record = its HW_ID register (which SIMD it sits on)
  int dbg_step = 0;
#ifdef TSH_PROBES
  const int dbg_slot = wave == 0 ? 0 : (wave == 4 ? 1 : (wave == 1 ? 2 : (wave == 5 ? 3 : -1)));
  const bool dbg_on = (a.dbg & 32) && blockIdx.x == 0 && dbg_slot >= 0;
  auto stamp = [&](int point) {
    if (dbg_on && dbg_step < 96 && lane == 0) a.dbg_buf[(dbg_slot * 96 + dbg_step) * 12 + point] = __builtin_amdgcn_s_memtime();
  };
  if (dbg_on && lane == 0) a.dbg_buf[(dbg_slot * 96) * 12 + 11] = (uint64_t)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
#else
  auto stamp = [](int) {};
#endif
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  using S3 = std::integral_constant<int, 3>;
  using No = std::false_type;
  using Yes = std::true_type;
  // LOAD(c): fragments of chunk c (stage CUR); my pieces of chunk c + 3 into the stage chunk c - 1 has left (both
  // groups are past their reads of it: the barrier before this phase); then my pieces of chunk c + 1 must have
  // landed (the 2 PPW younger ones may still fly) and my fragments must be in registers.
  auto load_phase = [&](auto CUR, auto FILL, auto SWITCH) {
    stamp(0);
    if (decltype(SWITCH)::value) next_tile_src();
    read_frags(CUR);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < PK; ++u) issue_one(FILL, u);
    f16_wait_dma<NPC + PK>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    stamp(5);
  };
  auto compute_phase = [&](auto FILL, auto LAST) {
    // the chunk's other PPW - PK pieces go out between the MFMAs, evenly spaced (an LDS-DMA instruction costs the
    // issuing wave ~60 cycles; behind an MFMA half of that is the matrix pipe's own time)
    constexpr int NM = 4 * MI, NC = NPC - PK, GAP = NC > 0 ? NM / NC : NM;
    int n = 0;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[sl][i], fb[sl][j], acc[i][j], 0, 0, 0);
          ++n;
          if (NC > 0 && n % GAP == GAP / 2 && n / GAP < NC) {
            __builtin_amdgcn_sched_barrier(0);
            issue_one(FILL, PK + n / GAP);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    __builtin_amdgcn_sched_barrier(0);
    stamp(6);
    // The phase's closing barrier.  After a tile's LAST chunk group 0 passes it before its epilogue (its partner is
    // waiting there to multiply the same chunk) and group 1 after its own (nothing of group 0's next LOAD depends
    // on it): the two epilogues of a SIMD then run side by side instead of one after the other.
    if (!decltype(LAST)::value || !wm) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    stamp(7);
    ++dbg_step;
  };

  // survivors whose list slots have been asked for but not yet answered (two per lane: F16_HITS = 128)
  uint2 pend_e[2] = {uint2{0u, 0u}, uint2{0u, 0u}};
  uint32_t pend_p[2] = {0u, 0u}, pend_n = 0;
  int pend_qbase = 0, pend_nbase = 0;
  auto ask = [&](int qb, int qrow) -> uint32_t {  // reserve one slot in the query's candidate list
    int q = qb + qrow;
    asm volatile("" : "+v"(q));  // keeps the list addresses of a patch from being precomputed (and spilled)
    return atomicAdd(&a.cand_cnt[(int64_t)q * CC_STRIDE], 1u);
  };
  auto store = [&](int qb, int nb, uint32_t e_key, uint32_t e_loc, uint32_t p) {
    if (p < (uint32_t)a.cand_cap) {
      const int64_t o = (int64_t)(qb + (int)(e_loc >> 16)) * a.cand_cap + p;
      const int pos = nb + (int)(e_loc & 0xFFFFu);
      a.cand_key[o] = e_key;
      a.cand_row[o] = a.row_ids ? a.row_ids[pos] : (uint32_t)pos;  // (a norm-grouped plane: the row this position holds)
    }
  };
  auto settle = [&]() {  // the answers have had a whole tile's time to arrive
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if ((uint32_t)(lane + 64 * h) < pend_n) store(pend_qbase, pend_nbase, pend_e[h].x, pend_e[h].y, pend_p[h]);
    pend_n = 0;
  };

  float *my_seed = s_seed[wave];
  float *my_aux = s_aux[wave];
  float *my_alpha = s_alpha[wave];
  uint2 *my_hits = s_hits[wave];
  // L2: c_v = |v|^2 / (2 s) of this lane's two corpus rows (column blocks j = 0, 1) of a tile; +inf past the last row
  // (nothing passes, and no read past the norms' end)
  // ROWW: nv = an upper bound of |v| of the same two rows (0 past the last row)
  float cv[2] = {0.f, 0.f}, nv[2] = {0.f, 0.f};
  const float half_over_s = 0.5f / a.dot_scale;  // a power of two: exact
  auto load_cv = [&](int tile) {
    int qt, nt;
    batch_tile_of(a, tile, &qt, &nt);
    const int nb = a.row0 + nt * TN;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = nb + wn * 64 + j * 32 + l31;
      const float sq = col < a.row1 ? a.sqnorm[col] : 0.f;
      if (L2) cv[j] = col < a.row1 ? sq * half_over_s : __builtin_inff();
      if (ROWV) nv[j] = batch_norm_up(sq);
    }
  };
  if (L2 || ROWV) load_cv((int)blockIdx.x);

  // ---- prologue: three chunks in flight, chunk 0 landed; group 1 starts one phase late ---------------------------
  set_src(i_tile);
  int cur_q_tile = -1;
  issue(S0{});  // KC is a multiple of four (host) and at least four: every tile starts in ring stage 0
  issue(S1{});
  issue(S2{});
  f16_wait_dma<2 * NPC>();
  __builtin_amdgcn_s_barrier();
  if (wm) __builtin_amdgcn_s_barrier();

  for (int tile = (int)blockIdx.x; tile < total_tiles; tile += G) {
    int n_tile, q_tile;
    batch_tile_of(a, tile, &q_tile, &n_tile);
    const int qbase = q_tile * TM, nbase = a.row0 + n_tile * TN;
    if (q_tile != cur_q_tile) {
      // This wave's thresholds: with gridDim.x a multiple of 8 q_tiles a workgroup keeps its query tile for all its
      // tiles, so this runs once (and in the ragged tail of the tile order).  (The compiler waits for these loads
      // with vmcnt(0), which also drains the DMA in flight: harmless here, not in the K loop.)
      for (int t = lane; t < QROWS; t += 64) {
        const int q = qbase + wm * QROWS + t;
        // key = -(dot_acc * scale) <= thr  <=>  dot_acc >= -thr / scale = theta; the accumulator starts at -theta
        // (power-of-two scale: exact).  Capped at the largest key any row can have: every row still passes, and
        // the chain's start stays within the error model.  Padding rows: -inf, nothing passes.
        float th = -__builtin_inff();
        if (L2) {
          // seed = -(|q|^2 - thr') / (2 s): one rounding (the subtraction); thr' kept for the survivors' keys
          float ax = 0.f;
          if (q < a.nq) {
            const float tc = SEEDED ? __builtin_fminf(a.thr[q], a.kmax[q]) : 0.f;
            if (SEEDED) th = -((a.qsq[q] - tc) * half_over_s);
            ax = SEEDED ? tc : a.qsq[q];
          }
          my_aux[t] = ax;
        } else if (SEEDED && q < a.nq) {
          th = __builtin_fminf(a.thr[q], a.kmax[q]) / a.dot_scale;
        }
        my_seed[t] = th;
        // alpha_q in accumulator units: key units / (2 s) for L2 (acc = (thr' - key) / (2 s)), / s for -dot keys
        if (ROWW) my_alpha[t] = (q < a.nq && a.alpha) ? a.alpha[q] * (L2 ? half_over_s : 1.0f / a.dot_scale) : 0.f;
        if (ROWV && DENSE) my_alpha[t] = (q < a.nq && a.alpha) ? a.alpha[q] : 0.f;  // (key units)
      }
      cur_q_tile = q_tile;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (a wave's own LDS traffic is in order; the compiler too)
    }
    if (SEEDED) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        f32x4 sd[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) sd[gq] = *reinterpret_cast<const f32x4 *>(&my_seed[i * 32 + 4 * half + 8 * gq]);
        f32x4 al[4];
        if (ROWW) {
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) al[gq] = *reinterpret_cast<const f32x4 *>(&my_alpha[i * 32 + 4 * half + 8 * gq]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float st0 = L2 ? sd[r >> 2][r & 3] - cv[j] : sd[r >> 2][r & 3];
            acc[i][j][r] = ROWW ? __builtin_fmaf(al[r >> 2][r & 3], nv[j], st0) : st0;
          }
      }
    } else {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- K loop: LOAD | barrier | COMPUTE | barrier, four chunks (one turn of the ring) per iteration ---------------
    for (int kq = 0; kq < KC / 4 - 1; ++kq) {
      load_phase(S0{}, S3{}, No{});
      compute_phase(S3{}, No{});
      load_phase(S1{}, S0{}, No{});
      compute_phase(S0{}, No{});
      load_phase(S2{}, S1{}, No{});
      compute_phase(S1{}, No{});
      load_phase(S3{}, S2{}, No{});
      compute_phase(S2{}, No{});
    }
    load_phase(S0{}, S3{}, No{});   // issues my last chunk
    compute_phase(S3{}, No{});
    load_phase(S1{}, S0{}, Yes{});  // ... and from here on the first three chunks of my next tile
    compute_phase(S0{}, No{});
    load_phase(S2{}, S1{}, No{});
    compute_phase(S1{}, No{});
    load_phase(S3{}, S2{}, No{});
    compute_phase(S2{}, Yes{});

    // ---- epilogue ----------------------------------------------------------------------------------------------
    // C/D map of the 32x32 MFMA: col = lane & 31 (corpus row), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (query).
#ifdef TSH_PROBES
    if (a.dbg & 4) {  // probe: no epilogue (results are wrong)
      if (wm) __builtin_amdgcn_s_barrier();
      continue;
    }
#endif
    stamp(8);  // (lands in the record of the NEXT chunk: dbg_step has moved on)
    if (!DENSE) settle();  // the previous tile's
    stamp(10);
    uint32_t n_hits = 0;   // wave-uniform
    // (opaque per tile: everything the epilogue derives from the lane id is invariant across the tiles of the
    // persistent loop, and the compiler would hoist the site constants out of it -- and spill them)
    int half_t = half, l31_t = l31;
    asm volatile("" : "+v"(half_t), "+v"(l31_t));
    int cl[2];
    bool col_ok[2], alive_l[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      cl[j] = wn * 64 + j * 32 + l31_t;  // row within the tile
      const int col = nbase + cl[j];
      col_ok[j] = col < a.row1;
      bool alive = col_ok[j];
      if (col_ok[j]) {
        const int idc = a.row_ids ? (int)a.row_ids[col] : col;  // (a gathered copy: the row this plane position holds)
        if (a.live) alive = (a.live[idc >> 6] >> (idc & 63)) & 1ull;
        if (alive && a.mask) alive = (a.mask[idc >> 6] >> (idc & 63)) & 1ull;
      }
      alive_l[j] = alive;
    }
    // L2, filtered pass: the NEXT tile's c_v, issued beside the liveness words above so that one wait covers both and
    // the rest of the epilogue hides it (this tile's went into the accumulators' start values and is not needed again)
    if ((L2 || ROWV) && !DENSE && tile + G < total_tiles) load_cv(tile + G);
    // one survivor: into the wave's LDS list (slot = running count + rank among the passing lanes; no atomics).  A list
    // that is full is handed over on the spot -- one round of slot requests for all its entries, two per lane -- and
    // starts again empty: a tile whose rows pass for a tenth of its queries (a corpus whose short rows sit together)
    // pays a round trip per 128 survivors, not one per append
    auto flush = [&]() {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if ((uint32_t)(lane + 64 * h) < n_hits) {
          const uint2 e = my_hits[lane + 64 * h];
          store(qbase, nbase, e.x, e.y, ask(qbase, (int)(e.y >> 16)));
        }
      n_hits = 0;
    };
    auto append = [&](uint64_t m, bool mine, float key, int qrow, int cj) {
      const uint32_t np = (uint32_t)__popcll(m);
      if (__builtin_expect(n_hits + np > (uint32_t)F16_HITS, 0)) flush();  // (wave-uniform)
      const uint32_t slot = n_hits + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      const uint32_t loc = ((uint32_t)qrow << 16) | (uint32_t)cj;
      const uint32_t spare = (uint32_t)F16_HITS + (uint32_t)lane;
      my_hits[mine ? slot : spare] = uint2{__float_as_uint(key), loc};
      n_hits += np;
    };
    float scale_w = a.dot_scale;
    asm volatile("" : "+s"(scale_w));
    // a survivor's key from its accumulator: IP / cosine -(dot * scale) = (seed - acc) * scale; L2 thr' - 2 s acc
    auto key_of = [&](int ql, float av) -> float {
      return L2 ? __builtin_fmaf(-2.f * scale_w, av, my_aux[ql]) : (my_seed[ql] - av) * scale_w;
    };
    // CODE SIZE matters here: the epilogue runs once per tile, and whatever of it is not in the instruction cache
    // comes from L2 a line at a time (a first version with the 2 MI blocks' register walks unrolled -- 70 KB of
    // code for the kernel -- spent 5-8 k cycles per wave and tile in here for ~500 executed instructions).
    auto block_of = [&](int b) -> f32x16 {  // (a block by run-time number: copied into one register set)
      f32x16 t;
      switch (b) {
        case 0: t = acc[0][0]; break;
        case 1: t = acc[0][1]; break;
        case 2: t = acc[1][0]; break;
        case 3: t = acc[1][1]; break;
        case 4: t = acc[MI - 2][0]; break;
        case 5: t = acc[MI - 2][1]; break;
        case 6: t = acc[MI - 1][0]; break;
        default: t = acc[MI - 1][1]; break;
      }
      return t;
    };
    if (DENSE) {
#pragma clang loop unroll(disable)
      for (int b = 0; b < 2 * MI; ++b) {
        const f32x16 t = block_of(b);
        const int i = b >> 1, j = b & 1;
        const int cj = j ? cl[1] : cl[0];
        const bool okj = j ? col_ok[1] : col_ok[0], alj = j ? alive_l[1] : alive_l[0];
        float *dst = a.dense + (int64_t)(qbase + wm * QROWS + i * 32 + 4 * half_t) * a.dense_ld + (nbase + cj - a.row0);
        const float sq = L2 ? (j ? cv[1] : cv[0]) * (2.f * scale_w) : 0.f;  // |v|^2 back from c_v (exact scaling)
        const float nvj = ROWV ? (j ? nv[1] : nv[0]) : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (okj) {
            const int ql = i * 32 + 4 * half_t + (r & 3) + 8 * (r >> 2);
            float key = -(t[r] * scale_w);
            if (L2) key = __builtin_fmaf(-2.f * scale_w, t[r], my_aux[ql] + sq);
            if (ROWV) key = __builtin_fmaf(my_alpha[ql], nvj, key);  // (alpha = 0 without per-row bands: the key itself)
            dst[(int64_t)((r & 3) + 8 * (r >> 2)) * a.dense_ld] = alj ? key : __builtin_nanf("");
          }
      }
    } else {
      // Three steps, the first two straight-line vector code (a ballot -> branch hop costs ~100 cycles; a loop over
      // registers with one per register took ~200 cycles per register):
      //  1. per 32 x 32 block: does it hold a survivor at all (dead rows included)?  -> hit_blocks (wave-uniform).
      //     "Some accumulator is >= 0" on the bit patterns: as signed integers the values with a clear sign bit are
      //     the non-negative ones, and v_max3_i32 needs no NaN quieting in front of it.  (-0.0 counts as negative
      //     here and below: an accumulator is -0.0 only if it started there -- a padding query.)
      //  2. per block that does (3.7 of eight at k = 100): the 16 sign bits of a lane side by side, one v_alignbit
      //     per register: ({nm, c} >> 31) = (nm << 1) | sign(c); bit 15 - r set = register r negative.  A lone
      //     survivor's value is the block maximum of step 1.  The notes accumulate over the MI blocks of a column
      //     block j (a lane = one corpus row of j) and hold a lane's first TWO survivors: with independent queries a
      //     fifth of all wave tiles has some lane whose row passes for two of the wave's 32 MI queries
      //  3. per column block: one append of the lanes that hold one, another of those that hold two; if some lane holds
      //     more, or two in one block (1 wave tile in 2700), the slow walk over the hit blocks, register by register,
      //     does the whole tile instead
      uint32_t hit_blocks = 0;
      int mblk[2 * MI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f32x16 &c = acc[i][j];
          auto bits = [&](int r) -> int { return (int)__float_as_uint(c[r]); };
          auto imax = [](int x, int y) -> int { return x > y ? x : y; };
          const int m0 = imax(imax(bits(0), bits(1)), bits(2)), m1 = imax(imax(bits(3), bits(4)), bits(5));
          const int m2 = imax(imax(bits(6), bits(7)), bits(8)), m3 = imax(imax(bits(9), bits(10)), bits(11));
          const int m4 = imax(imax(bits(12), bits(13)), bits(14));
          mblk[2 * i + j] = imax(imax(imax(m0, m1), m2), imax(imax(m3, m4), bits(15)));
          if (__ballot(mblk[2 * i + j] >= 0) != 0) hit_blocks |= 1u << (2 * i + j);
        }
      }
      stamp(11);
      int cnt[2] = {0, 0}, rr[2][2] = {{0, 0}, {0, 0}};  // per column block: passing registers; the first two: 16 i + r
      float val[2][2] = {{0.f, 0.f}, {0.f, 0.f}};        // ... and their accumulators
      bool crowded = false;  // a lane holds two survivors inside ONE block: the block maximum is not their value
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (hit_blocks & (1u << (2 * i + j))) {  // wave-uniform
            const f32x16 &c = acc[i][j];
            uint32_t nm = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) nm = __builtin_amdgcn_alignbit(nm, __float_as_uint(c[r]), 31);
            const uint32_t pm = ~nm & 0xFFFFu;
            const int np = __builtin_popcount(pm);
            const bool has = np != 0, first = cnt[j] == 0;
            const float v = __uint_as_float((uint32_t)mblk[2 * i + j]);
            const int w = 16 * i + 15 - (int)__builtin_ctz(pm | 0x10000u);
            val[j][0] = has && first ? v : val[j][0];
            rr[j][0] = has && first ? w : rr[j][0];
            val[j][1] = has && !first ? v : val[j][1];
            rr[j][1] = has && !first ? w : rr[j][1];
            crowded |= np > 1;
            cnt[j] += np;
          }
        }
      }
      stamp(1);
#ifdef TSH_PROBES
      if ((a.dbg & 32) && a.dbg_buf && lane == 0) {  // how often the register-by-register walk runs, and why
        unsigned long long *cb = reinterpret_cast<unsigned long long *>(a.dbg_buf) + 4 * 96 * 12;
        atomicAdd(cb + 0, 1ull);
        atomicAdd(cb + 3, (unsigned long long)__builtin_popcount(hit_blocks));
      }
      {
        const bool c2 = __ballot((alive_l[0] && crowded) || (alive_l[1] && crowded)) != 0;
        const bool m2 = __ballot((alive_l[0] && cnt[0] > 2) || (alive_l[1] && cnt[1] > 2)) != 0;
        if ((a.dbg & 32) && a.dbg_buf && lane == 0) {
          unsigned long long *cb = reinterpret_cast<unsigned long long *>(a.dbg_buf) + 4 * 96 * 12;
          if (c2) atomicAdd(cb + 1, 1ull);
          if (m2) atomicAdd(cb + 2, 1ull);
        }
      }
#endif
      if (__builtin_expect(__ballot((alive_l[0] && (cnt[0] > 2 || crowded)) || (alive_l[1] && (cnt[1] > 2 || crowded))) != 0, 0)) {
        while (hit_blocks) {
          const int b = __builtin_ctz(hit_blocks);
          hit_blocks &= hit_blocks - 1;
          const f32x16 t = block_of(b);
          const int i = b >> 1, j = b & 1;
          const bool alj = j ? alive_l[1] : alive_l[0];
          const int cj = j ? cl[1] : cl[0];
          const int rloc = i * 32 + 4 * half_t;
          // which of the block's 16 registers hold a survivor in SOME live lane: the lanes' sign masks (one v_alignbit
          // per register, as in step 2) and sixteen ballots in a row, no branch between them -- then only those
          // registers are walked (one or two of the sixteen, usually).  A corpus whose survivors pile up on a few rows
          // (L2 / IP with varying norms: the short rows are near EVERY query) takes this path for a good part of its
          // tiles; with a ballot and a branch per register it cost such a corpus 20 % of its key passes.
          uint32_t nm = 0;
#pragma unroll
          for (int r = 0; r < 16; ++r) nm = __builtin_amdgcn_alignbit(nm, __float_as_uint(t[r]), 31);
          const uint32_t pm = alj ? (~nm & 0xFFFFu) : 0u;  // bit 15 - r = register r passes in this lane
          uint32_t any = 0;
#pragma unroll
          for (int r = 0; r < 16; ++r) any |= __ballot((pm >> (15 - r)) & 1u) != 0 ? (1u << r) : 0u;
          while (any) {
            const int r = __builtin_ctz(any);
            any &= any - 1;
            const int ql = rloc + (r & 3) + 8 * (r >> 2);
            const float av = t[r];  // (r is wave-uniform: an indexed register read)
            const bool mine = ((pm >> (15 - r)) & 1u) != 0;
            const uint64_t mm = __ballot(mine);
            append(mm, mine, key_of(ql, av), wm * QROWS + ql, cj);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int sl = 0; sl < 2; ++sl) {
            const bool mine = alive_l[j] && cnt[j] > sl;
            const uint64_t mm = __ballot(mine);
            if (mm) {
              const int w = rr[j][sl];
              const int ql = (w >> 4) * 32 + 4 * half_t + (w & 3) + 8 * ((w & 15) >> 2);
              append(mm, mine, key_of(ql, val[j][sl]), wm * QROWS + ql, cl[j]);
            }
          }
        }
      }
      stamp(4);
    }
    if (!DENSE) {
      const uint32_t listed = n_hits;  // (<= F16_HITS: append)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if ((uint32_t)(lane + 64 * h) < listed) {
          pend_e[h] = my_hits[lane + 64 * h];
          pend_p[h] = ask(qbase, (int)(pend_e[h].y >> 16));
        }
      pend_n = listed;
      pend_qbase = qbase;
      pend_nbase = nbase;
    }
    if ((L2 || ROWV) && DENSE && tile + G < total_tiles) load_cv(tile + G);  // (the dense keys above still needed this tile's)
    stamp(9);
    __builtin_amdgcn_sched_barrier(0);
    if (wm) __builtin_amdgcn_s_barrier();  // (group 1's barrier of the tile's last COMPUTE phase: see compute_phase)
    __builtin_amdgcn_sched_barrier(0);
  }
  if (!wm) __builtin_amdgcn_s_barrier();  // group 1's last phase
  if (!DENSE) settle();
  f16_wait_dma<0>();  // the stream's surplus pieces must have landed before this workgroup's LDS is handed on
#ifdef TSH_PROBES
  if (dbg_sink[0] == 0x12345678u && a.dbg_buf) a.dbg_buf[0] = 1;  // (keeps the sink alive)
#endif
}

}  // namespace tsh
