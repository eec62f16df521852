// tsh_scan_tu.hip -- the single-query scan kernels' instantiations and their launcher (K1, tsh_kernels.hip.h).
// A translation unit of its own: the instantiations compile in parallel with the rest of the library.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>

#include "../../include/tostore_hip.h"
#include "tsh_launch.h"

namespace tsh {
namespace {

// (rows per group, min waves per SIMD) per row width: two register buffers of
// R*NCH*4 VGPRs plus NCH*4 for the query must fit 512/MINW registers.
template <int NCH, bool MASKED> struct ScanTune {
  static constexpr int R = (NCH <= 2) ? 4 : (NCH == 3 ? (MASKED ? 2 : 4) : 2);
  static constexpr int MINW =
      (NCH <= 2) ? 4 : (NCH == 3 ? (MASKED ? 4 : 3) : (NCH == 4 ? 4 : (NCH <= 6 ? 3 : (NCH <= 8 ? 2 : 1))));
};
#define TSH_LAUNCH(KERN, GRID, BLOCK, LDS, ST, EV, ARG)                                                  \
  do {                                                                                                  \
    if ((EV).start || (EV).stop)                                                                        \
      hipExtLaunchKernelGGL(KERN, dim3((unsigned)(GRID)), dim3((unsigned)(BLOCK)), (unsigned)(LDS), ST, \
                            (EV).start, (EV).stop, 0, ARG);                                             \
    else                                                                                                \
      KERN<<<(GRID), (BLOCK), (unsigned)(LDS), ST>>>(ARG);                                              \
  } while (0)

// How many waves of a dense scan share a CU.  The registers would allow 8 to 16, and that is slower: every wave
// streams its own tile, and HBM serves fewer, longer streams better than many short ones -- measured per width with
// the occupancy held down by an (unused) dynamic LDS allocation, two-wave workgroups (tools/dims_probe.sh; scan time
// at the default occupancy -> at the one chosen here): d = 768 457 -> 439 us (0.840 -> 0.875 of the HBM peak), 1536
// 562 -> 537, 2048 740 -> 705, 1000 367 -> 357, 512 191 -> 184, 384 147 -> 139.  Four waves per CU from three chunks
// per row on, eight for two; one-chunk rows (d <= 256) keep the registers' occupancy (d = 200: 76 us against 79 / 113
// at eight / four waves); two waves per CU are as good as four at d = 768 and one is far too few (657 us).  Masked
// scans that keep most rows (tombstones, mild filters: scan_mostly_live) stream like dense ones and take the same shape
// (keep 100 / 95 / 80 %: 481 / 460 / 385 -> 441-454 / 426-434 / 370 us); selective ones walk scattered rows and live on many
// waves (keep 50 %: +2 % at eight waves, keep 10 %: -8 %): theirs stays.  One
// tile per wave and a workgroup per two tiles stays: a grid of 480 / 960 / 1920 workgroups striding over the tiles
// takes 560 / 500 / 457 us against 439-445.
template <int NCH> struct ScanShape {
  static constexpr int WPB = NCH == 1 ? 4 : 2;                          // waves per workgroup
  static constexpr int LDS = NCH == 1 ? 0 : (NCH == 2 ? 32768 : 65536);  // 160 KB per CU: 4 resp. 2 workgroups
};

template <int NCH, int METRIC, bool FULL, bool MASKED>
void launch_scan_t(const ScanArgsQ &a, int grid, hipStream_t s, const LaunchEv &ev, bool mostly_live) {
  using T = ScanTune<NCH, MASKED>;
  // grid > 0: a big shard -- dense scans in the shape of ScanShape, masked ones in 4-wave workgroups;
  // grid < 0: -grid one-wave workgroups (small shards)
  using S = ScanShape<NCH>;
  if (grid > 0 && (!MASKED || mostly_live))
    TSH_LAUNCH((scan_kernel<NCH, METRIC, FULL, MASKED, T::R, true, 4, T::MINW>), (a.a.n_tiles + S::WPB - 1) / S::WPB,
               64 * S::WPB, S::LDS, s, ev, a);
  else if (grid > 0) TSH_LAUNCH((scan_kernel<NCH, METRIC, FULL, MASKED, T::R, true, 4, T::MINW>), grid, 256, 0, s, ev, a);
  else TSH_LAUNCH((scan_kernel<NCH, METRIC, FULL, MASKED, T::R, true, 4, T::MINW>), -grid, 64, 0, s, ev, a);
}
template <int NCH, int METRIC>
void launch_scan_m(const ScanArgsQ &a, bool masked, int grid, hipStream_t s, const LaunchEv &ev, bool ml) {
  bool full = a.a.d4 == NCH * 64;
  if (full) {
    if (masked) launch_scan_t<NCH, METRIC, true, true>(a, grid, s, ev, ml);
    else launch_scan_t<NCH, METRIC, true, false>(a, grid, s, ev, ml);
  } else {
    if (masked) launch_scan_t<NCH, METRIC, false, true>(a, grid, s, ev, ml);
    else launch_scan_t<NCH, METRIC, false, false>(a, grid, s, ev, ml);
  }
}
template <int NCH>
void launch_scan_n(const ScanArgsQ &a, int metric, bool masked, int grid, hipStream_t s, const LaunchEv &ev, bool ml) {
  if (metric == TSH_METRIC_L2) launch_scan_m<NCH, METRIC_L2>(a, masked, grid, s, ev, ml);
  else if (metric == TSH_METRIC_IP) launch_scan_m<NCH, METRIC_IP>(a, masked, grid, s, ev, ml);
  else launch_scan_m<NCH, METRIC_COS>(a, masked, grid, s, ev, ml);
}
template <int SPLIT>
void launch_packed(const ScanArgsQ &a, int metric, bool masked, int grid, int threads, hipStream_t s,
                   const LaunchEv &ev) {
#define TSH_PK(M, MK) TSH_LAUNCH((scan_packed_kernel<SPLIT, M, MK, true>), grid, threads, 0, s, ev, a)
  if (metric == TSH_METRIC_L2) { if (masked) TSH_PK(METRIC_L2, true); else TSH_PK(METRIC_L2, false); }
  else if (metric == TSH_METRIC_IP) { if (masked) TSH_PK(METRIC_IP, true); else TSH_PK(METRIC_IP, false); }
  else { if (masked) TSH_PK(METRIC_COS, true); else TSH_PK(METRIC_COS, false); }
#undef TSH_PK
}

template <int NCH>
void launch_scan_list_n(const ScanArgsQ &a, int metric, hipStream_t s, const LaunchEv &ev) {
  constexpr int R = NCH <= 5 ? 4 : 2;  // two register buffers of R rows: all eight rows of a wave in flight up to d = 1280
  const int grid = a.a.n_tiles;
  if (metric == TSH_METRIC_L2) TSH_LAUNCH((scan_list_kernel<NCH, METRIC_L2, R, true>), grid, 512, 0, s, ev, a);
  else if (metric == TSH_METRIC_IP) TSH_LAUNCH((scan_list_kernel<NCH, METRIC_IP, R, true>), grid, 512, 0, s, ev, a);
  else TSH_LAUNCH((scan_list_kernel<NCH, METRIC_COS, R, true>), grid, 512, 0, s, ev, a);
}

}  // namespace

bool scan_list_supported(int nch, int64_t ld) { return nch >= 1 && nch <= 8 && ld != 128 && ld != 64 && ld != 32; }

void launch_scan_list(const ScanArgsQ &a, int nch, int metric, hipStream_t s, const LaunchEv &ev) {
  if (a.a.n_tiles <= 0) return;
  switch (nch) {
    case 1: launch_scan_list_n<1>(a, metric, s, ev); break;
    case 2: launch_scan_list_n<2>(a, metric, s, ev); break;
    case 3: launch_scan_list_n<3>(a, metric, s, ev); break;
    case 4: launch_scan_list_n<4>(a, metric, s, ev); break;
    case 5: launch_scan_list_n<5>(a, metric, s, ev); break;
    case 6: launch_scan_list_n<6>(a, metric, s, ev); break;
    case 7: launch_scan_list_n<7>(a, metric, s, ev); break;
    default: launch_scan_list_n<8>(a, metric, s, ev); break;
  }
}

void launch_scan(const ScanArgsQ &a, int nch, int metric, bool masked, hipStream_t s, const LaunchEv &ev, bool ml) {
  int grid = (a.a.n_tiles + 3) / 4;
  if (grid < 1) grid = 1;
  if (a.a.ld == 128 || a.a.ld == 64 || a.a.ld == 32) {
    // narrow rows: several whole rows per 1 KiB wave load (scan_packed_kernel)
    int threads = 256;
    if (a.a.n_tiles < SMALL_SHARD_TILES) {
      grid = std::max(1, (int)a.a.n_tiles);
      threads = 64;
    }
    if (a.a.ld == 128) launch_packed<1>(a, metric, masked, grid, threads, s, ev);
    else if (a.a.ld == 64) launch_packed<2>(a, metric, masked, grid, threads, s, ev);
    else launch_packed<3>(a, metric, masked, grid, threads, s, ev);
    return;
  }
  // fewer than ~6 four-wave workgroups per CU: tile counts per CU differ by tens of
  // percent; one tile per workgroup lets the dispatcher even them out
  if (a.a.n_tiles < SMALL_SHARD_TILES) grid = -std::max(1, (int)a.a.n_tiles);
  switch (nch) {
    case 1: launch_scan_n<1>(a, metric, masked, grid, s, ev, ml); break;
    case 2: launch_scan_n<2>(a, metric, masked, grid, s, ev, ml); break;
    case 3: launch_scan_n<3>(a, metric, masked, grid, s, ev, ml); break;
    case 4: launch_scan_n<4>(a, metric, masked, grid, s, ev, ml); break;
    case 5: launch_scan_n<5>(a, metric, masked, grid, s, ev, ml); break;
    case 6: launch_scan_n<6>(a, metric, masked, grid, s, ev, ml); break;
    case 7: launch_scan_n<7>(a, metric, masked, grid, s, ev, ml); break;
    case 8: launch_scan_n<8>(a, metric, masked, grid, s, ev, ml); break;
    case 10: launch_scan_n<10>(a, metric, masked, grid, s, ev, ml); break;
    case 12: launch_scan_n<12>(a, metric, masked, grid, s, ev, ml); break;
    case 14: launch_scan_n<14>(a, metric, masked, grid, s, ev, ml); break;
    default: launch_scan_n<16>(a, metric, masked, grid, s, ev, ml); break;
  }
}

}  // namespace tsh
