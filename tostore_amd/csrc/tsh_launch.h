// tsh_launch.h -- launchers that live in translation units of their own (the kernels they instantiate take most
// of the library's compile time: tsh_scan_tu.hip, tsh_batch_tu.hip), as seen by the host side in tsh_lib.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "tsh_batch.hip.h"
#include "tsh_kernels.hip.h"

namespace tsh {

constexpr int SMALL_SHARD_TILES = 6 * 4 * 256;  // below this: one-wave workgroups, two scan streams

// Events that ride on the scan's own dispatch packet (hipExtLaunchKernel): a separate
// hipEventRecord is a barrier packet of its own, and two or three of those between
// consecutive scans were most of the gap between them.
struct LaunchEv {
  hipEvent_t start = nullptr, stop = nullptr;
};

// K1 for one query: picks the instantiation for the row width (nch), metric and mask, the grid shape for the
// shard size.  tsh_scan_tu.hip
// mostly_live: a masked scan that keeps most rows (tombstones, a mild filter) streams like a dense one and is launched
// in the dense scans' shape.
void launch_scan(const ScanArgsQ &a, int nch, int metric, bool masked, hipStream_t s, const LaunchEv &ev = LaunchEv(),
                 bool mostly_live = false);
// (live rows of a masked scan, as far as the host knows them) -> mostly_live
inline bool scan_mostly_live(int64_t live_rows, int64_t rows) { return live_rows * 10 >= rows * 6; }

// K1 over a compacted list of row ids (selective masks: a.a.list, a.a.n_tiles = list tiles).  Row widths up to eight
// 1 KiB chunks (d <= 2048) that are not served by the packed kernels; tsh_scan_tu.hip
bool scan_list_supported(int nch, int64_t ld);
void launch_scan_list(const ScanArgsQ &a, int nch, int metric, hipStream_t s, const LaunchEv &ev = LaunchEv());

// batched key pass (f32 MFMA / bf16x3 / f16 by a.Vs and a.dot_scale).  cus: compute units of the device the stream
// belongs to (the persistent f16 kernels run one workgroup per CU).  tsh_batch_tu.hip
void launch_batch_score_m(int metric, const BatchArgs &a, bool dense, hipStream_t st, int cus);

}  // namespace tsh
