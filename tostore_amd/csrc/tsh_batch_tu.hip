// tsh_batch_tu.hip -- the batched path's key kernels (f32 MFMA, bf16x3, f16: tsh_batch.hip.h, tsh_batch_f16.hip.h)
// and their launcher.  A translation unit of its own: these kernels are the ones that get tuned, and they
// compile in parallel with the rest of the library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "../../include/tostore_hip.h"
#include "tsh_launch.h"
#include "tsh_batch_f16.hip.h"
#include "tsh_batch_f16pp.hip.h"

namespace tsh {
namespace {

template <int METRIC>
void launch_batch_score(const BatchArgs &a, bool dense, hipStream_t st) {
  int grid = a.q_tiles * a.n_tiles;
  if (grid <= 0) return;
  if (dense) batch_score_kernel<METRIC, true><<<grid, BT_THREADS, 0, st>>>(a);
  else batch_score_kernel<METRIC, false><<<grid, BT_THREADS, 0, st>>>(a);
}
template <int METRIC>
void launch_batch_score_bf16(const BatchArgs &a, bool dense, hipStream_t st, int cus) {
  int grid = a.q_tiles * a.n_tiles;
  if (grid <= 0) return;
  if (a.dot_scale != 0.f) {  // f16 variant: tsh_batch_f16.hip.h (persistent: one 8-wave workgroup per CU)
    const int pgrid = std::min(grid, cus > 0 ? cus : 256);  // (cus: the launching shard's device, DeviceStreams::cus)
    // Third generation (ping-pong phases, tsh_batch_f16pp.hip.h) for every metric; L2's per-row term rides in the
    // accumulators' start values since round 4.  Probe builds (-DTSH_PROBES) carry the second generation beside it,
    // TSH_F16_GEN=2 selects it there (tools/r4_ab_l2.sh)
#ifdef TSH_PROBES
    static const bool gen2 = getenv("TSH_F16_GEN") != nullptr && getenv("TSH_F16_GEN")[0] == '2';
    if (gen2) {
      if (a.tile_m == 256) {
        if (dense) batch_score_f16_kernel<METRIC, true, 4><<<pgrid, 512, 0, st>>>(a);
        else batch_score_f16_kernel<METRIC, false, 4><<<pgrid, 512, 0, st>>>(a);
      } else {
        if (dense) batch_score_f16_kernel<METRIC, true, 2><<<pgrid, 512, 0, st>>>(a);
        else batch_score_f16_kernel<METRIC, false, 2><<<pgrid, 512, 0, st>>>(a);
      }
      return;
    }
#endif
    if (a.tile_m == 256) {
      if (dense) batch_score_f16pp_kernel<METRIC, true, 4><<<pgrid, 512, 0, st>>>(a);
      else batch_score_f16pp_kernel<METRIC, false, 4><<<pgrid, 512, 0, st>>>(a);
    } else {
      if (dense) batch_score_f16pp_kernel<METRIC, true, 2><<<pgrid, 512, 0, st>>>(a);
      else batch_score_f16pp_kernel<METRIC, false, 2><<<pgrid, 512, 0, st>>>(a);
    }
    return;
  }
  if (a.tile_m == 256) {  // 256 x 256 tiles, 8 waves (batches of more than 128 queries)
    if (dense) batch_score_bf16x3_kernel<METRIC, true, 256, 256, 128><<<grid, 512, 0, st>>>(a);
    else batch_score_bf16x3_kernel<METRIC, false, 256, 256, 128><<<grid, 512, 0, st>>>(a);
    return;
  }
  if (dense) batch_score_bf16x3_kernel<METRIC, true><<<grid, BT_THREADS, 0, st>>>(a);
  else batch_score_bf16x3_kernel<METRIC, false><<<grid, BT_THREADS, 0, st>>>(a);
}
}  // namespace

void launch_batch_score_m(int metric, const BatchArgs &a, bool dense, hipStream_t st, int cus) {
  if (a.Vs) {
    if (metric == TSH_METRIC_L2) launch_batch_score_bf16<METRIC_L2>(a, dense, st, cus);
    else if (metric == TSH_METRIC_IP) launch_batch_score_bf16<METRIC_IP>(a, dense, st, cus);
    else launch_batch_score_bf16<METRIC_COS>(a, dense, st, cus);
    return;
  }
  if (metric == TSH_METRIC_L2) launch_batch_score<METRIC_L2>(a, dense, st);
  else if (metric == TSH_METRIC_IP) launch_batch_score<METRIC_IP>(a, dense, st);
  else launch_batch_score<METRIC_COS>(a, dense, st);
}

}  // namespace tsh
