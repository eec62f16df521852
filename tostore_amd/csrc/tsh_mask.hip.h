// tsh_mask.hip.h -- a row mask that lives on the device: the kept rows of a selective mask as an ascending list of
// row ids, made ON the device (gfx950, wave = 64).
//
// The pointer form of a masked search (tsh_search's row_mask) slices the caller's bitmap, counts it and lists its rows
// on the host, call by call (tsh_host_sync.h list_mask_bits: 43 + 10 us for a 1 M-row mask at 1 % on a 2.1 GHz core --
// more than half of what a lone masked query takes).  A mask HANDLE (tsh_mask_create, include/tostore_hip.h) is sliced
// once; its words go to the device once; and the list the selective scans read (scan_list_kernel,
// exact_scan_kernel: tsh_kernels.hip.h, tsh_exact.hip.h) is compacted here, by two small launches:
//   M1 mask_block_count_kernel   one workgroup per 256 words (16 384 rows): popcount of its words -> bsum[block]
//   M2 mask_compact_kernel       same grid: a workgroup's first list position = the sum of bsum[0 .. block) (every
//                                thread adds a few of them: 62 blocks at 1 M rows), a thread's = that + the exclusive
//                                prefix of the popcounts in front of it (wave scan by DPP shuffles + the four wave
//                                totals through LDS); it then writes its word's set bits as row ids, ascending.
// The list's padding to whole 64-entry tiles (0xFFFFFFFF) is a memset behind M2.  No reference counterpart: the
// reference's vectorSearch takes no filter; the row sets this serves are the ones that live across
// queries there -- tombstones (/root/reference/lib/src/core/ngh_page.dart:105-108) and a WHERE's primary keys mapped
// through the pk -> nodeId tree (/root/reference/lib/src/core/vector_index_manager.dart:1223-1378).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace tsh {

constexpr int MASK_BLOCK_WORDS = 256;  // words (of 64 rows) per workgroup of M1 / M2: one per thread

__device__ __forceinline__ uint32_t mask_wave_sum(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
  return v;
}

// M1: bsum[b] = kept rows of words [256 b, 256 b + 256)
static __global__ void __launch_bounds__(MASK_BLOCK_WORDS) mask_block_count_kernel(const uint64_t *__restrict__ words,
                                                                                   int32_t n_words,
                                                                                   uint32_t *__restrict__ bsum) {
  __shared__ uint32_t s_w[MASK_BLOCK_WORDS / 64];
  const int w = blockIdx.x * MASK_BLOCK_WORDS + threadIdx.x;
  const uint32_t c = mask_wave_sum(w < n_words ? (uint32_t)__popcll(words[w]) : 0u);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < MASK_BLOCK_WORDS / 64; ++i) t += s_w[i];
    bsum[blockIdx.x] = t;
  }
}

// M2: list[...] = ids of the set bits, ascending; *total = their number (written by the last workgroup)
static __global__ void __launch_bounds__(MASK_BLOCK_WORDS) mask_compact_kernel(const uint64_t *__restrict__ words,
                                                                               int32_t n_words,
                                                                               const uint32_t *__restrict__ bsum,
                                                                               uint32_t *__restrict__ list,
                                                                               uint32_t *__restrict__ total) {
  __shared__ uint32_t s_red[MASK_BLOCK_WORDS / 64], s_tot[MASK_BLOCK_WORDS / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // where this workgroup's rows start in the list
  uint32_t part = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += MASK_BLOCK_WORDS) part += bsum[i];
  part = mask_wave_sum(part);
  const int w = blockIdx.x * MASK_BLOCK_WORDS + threadIdx.x;
  uint64_t word = w < n_words ? words[w] : 0ull;
  const uint32_t c = (uint32_t)__popcll(word);
  uint32_t incl = c;  // inclusive prefix within the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t u = (uint32_t)__shfl_up((int)incl, d);
    if (lane >= d) incl += u;
  }
  if (lane == 0) s_red[wave] = part;
  if (lane == 63) s_tot[wave] = incl;
  __syncthreads();
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < MASK_BLOCK_WORDS / 64; ++i) {
    o += s_red[i];                 // the workgroup's base, in four parts
    if (i < wave) o += s_tot[i];   // the waves in front of this one
  }
  o += incl - c;
  const uint32_t base = (uint32_t)w * 64u;
  for (; word; word &= word - 1) list[o++] = base + (uint32_t)__builtin_ctzll(word);
  if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == MASK_BLOCK_WORDS - 1) *total = o;  // (the last thread's end)
}

}  // namespace tsh
