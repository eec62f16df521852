// tsh_lib.hip -- host side of libtostore_hip.so (C-ABI in include/tostore_hip.h).
//
// Owns the device-resident copy of one vector index's float32 embedding column
// (the data the reference keeps in <index>/ngh/rawvec pages,
// /root/reference/lib/src/core/ngh_page.dart:310-450) and answers
// NghGraphEngine.search-shaped queries (ngh_graph_engine.dart:67-135) with the
// kernels in tsh_kernels.hip.h.  No CPU fallback exists: without a device every
// compute entry returns TSH_E_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cctype>
#include <chrono>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <unistd.h>

#include "../../include/tostore_hip.h"
#include "tsh_batch.hip.h"
#include "tsh_batch_f16.hip.h"
#include "tsh_exact.hip.h"
#include "tsh_host_sync.h"
#include "tsh_kernels.hip.h"
#include "tsh_launch.h"
#include "tsh_mask.hip.h"
#include "tsh_pq.hip.h"

using namespace tsh;

namespace {

thread_local std::string g_err;

int set_err(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(expr)                                                                  \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess)                                                             \
      return set_err(e_ == hipErrorOutOfMemory ? TSH_E_OOM : TSH_E_HIP, "%s failed: %s", \
                     #expr, hipGetErrorString(e_));                                   \
  } while (0)

int device_count_cached() {
  static int n = [] {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
    return c;
  }();
  return n;
}

constexpr int MAX_CTX = 8;            // contexts (queries in flight) per shard
constexpr int PIPE_DEPTH = 8;         // queries a multi-query call keeps in flight
constexpr int F16_DENIAL_CALLS = 256; // batched calls a shard stays off fp16 keys after its lists kept overflowing
constexpr uint32_t QUARANTINE_MAX = 1024;  // quarantined rows per shard; beyond that the shard goes to safe mode
constexpr int SUBMIT_THREADS = 1;     // host threads that submit a multi-query call (more did not help: the pipeline is GPU-bound)
constexpr int MAX_DIM_SCAN = 4096;    // register-resident query (NCH <= 16; the reference's f32 pages hold d <= 4073)
constexpr float BIG_ABS = 1.0e15f;    // beyond this f32 squares can overflow

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// ---- Dart double.compareTo ([external] Dart SDK): NaN greatest (and equal to itself), -0 < +0 ------
// double.compareTo as an integer order: key(a) < key(b)  <=>  dart_compare(a, b) < 0, equal keys <=> compareTo == 0
// (every NaN maps to the one largest key; -0.0 sorts just below +0.0).  Sorting (key, id) pairs with integer
// compares is what the finaliser does for every query.
inline uint64_t dart_order_key(double d) {
  if (d != d) return ~0ull;
  uint64_t b;
  memcpy(&b, &d, 8);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
struct Hit {
  uint64_t key;  // dart_order_key(distance)
  int64_t id;
};
inline bool hit_less(const Hit &a, const Hit &b) { return a.key != b.key ? a.key < b.key : a.id < b.id; }
// Ascending by (key, id).  A query's candidates are k plus a band's worth of rows whose distances share their
// leading bits, and std::sort spends 3-4 us on 127 of them (two thirds of a query's finalisation, mostly
// mispredicted branches): one counting pass over 512 buckets of the key RANGE (a shift, so the bucket is monotone in
// the key) leaves runs of a few entries to put in order -- 1 us.  Ties, NaNs or a wild range only mean longer runs.
inline void sort_hits(Hit *h, size_t n) {
  constexpr int BITS = 9, NB = 1 << BITS;
  if (n < 24 || n > 60000) {
    std::sort(h, h + n, hit_less);
    return;
  }
  static thread_local std::vector<Hit> tmp;
  static thread_local std::vector<uint16_t> bucket;
  if (tmp.size() < n) {
    tmp.resize(n);
    bucket.resize(n);
  }
  uint64_t kmin = ~0ull, kmax = 0;
  for (size_t i = 0; i < n; ++i) {
    kmin = std::min(kmin, h[i].key);
    kmax = std::max(kmax, h[i].key);
  }
  const uint64_t range = kmax - kmin;
  const int shift = range < (uint64_t)NB ? 0 : (64 - __builtin_clzll(range)) - BITS;  // (range >> shift) < NB
  uint16_t pos[NB + 1];
  memset(pos, 0, sizeof(pos));
  for (size_t i = 0; i < n; ++i) {
    const uint16_t b = (uint16_t)((h[i].key - kmin) >> shift);
    bucket[i] = b;
    pos[b + 1]++;
  }
  for (int b = 0; b < NB; ++b) pos[b + 1] += pos[b];
  for (size_t i = 0; i < n; ++i) tmp[pos[bucket[i]]++] = h[i];
  size_t start = 0;
  for (size_t i = 1; i <= n; ++i) {
    if (i < n && ((tmp[i].key - kmin) >> shift) == ((tmp[start].key - kmin) >> shift)) continue;
    const size_t len = i - start;  // a bucket's run
    if (len > 12) {
      std::sort(tmp.begin() + start, tmp.begin() + i, hit_less);
    } else {
      for (size_t a = start + 1; a < i; ++a) {
        const Hit x = tmp[a];
        size_t j = a;
        for (; j > start && hit_less(x, tmp[j - 1]); --j) tmp[j] = tmp[j - 1];
        tmp[j] = x;
      }
    }
    start = i;
  }
  memcpy(h, tmp.data(), n * sizeof(Hit));
}
// the distance an order key stands for (every NaN comes back as the one quiet NaN)
inline double dart_order_key_to_double(uint64_t key) {
  if (key == ~0ull) return std::nan("");
  const uint64_t b = (key >> 63) ? (key & 0x7FFFFFFFFFFFFFFFull) : ~key;
  double d;
  memcpy(&d, &b, 8);
  return d;
}

// Final per-candidate arithmetic of ngh_graph_engine.dart:908-946 given the exact f64 sums (L2 :926 sqrt(s0);
// IP :914 -s0; cosine :944-945,:916 1 - s0 / (sqrt(mag_a) * sqrt(s1)), similarity 0 when the denominator is not
// positive), mag_a = sum q[i]*q[i] accumulated in element order --
// over a whole list: a plain loop over arrays, so the compiler's vector square roots and
// divisions (IEEE-exact, like the scalar ones; no contraction: -ffp-contract=off) do four or eight at a time on
// hosts that have AVX2 / AVX-512 -- the square root and the division were a third of a query's finalisation.
#define TSH_FINAL_KEYS_BODY                                                                   \
  for (uint32_t i = 0; i < n; ++i) {                                                          \
    const double s0 = e[i].s0, s1 = e[i].s1;                                                  \
    double d;                                                                                 \
    if (metric == TSH_METRIC_L2) {                                                            \
      d = std::sqrt(s0);                                                                      \
    } else if (metric == TSH_METRIC_IP) {                                                     \
      d = -s0;                                                                                \
    } else {                                                                                  \
      const double denom = sqrt_mag_a * std::sqrt(s1);                                        \
      const double sim = denom > 0 ? s0 / denom : 0;                                          \
      d = 1.0 - sim;                                                                          \
    }                                                                                         \
    dist[i] = d;                                                                              \
  }
#if defined(__x86_64__)
__attribute__((target("avx2"))) void final_distances_avx2(int metric, const BlockEntry *e, uint32_t n, double sqrt_mag_a,
                                                          double *dist) {
  TSH_FINAL_KEYS_BODY
}
#endif
void final_distances_base(int metric, const BlockEntry *e, uint32_t n, double sqrt_mag_a, double *dist) {
  TSH_FINAL_KEYS_BODY
}
#undef TSH_FINAL_KEYS_BODY
inline void final_distances(int metric, const BlockEntry *e, uint32_t n, double sqrt_mag_a, double *dist) {
#if defined(__x86_64__)
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2) return final_distances_avx2(metric, e, n, sqrt_mag_a, dist);
#endif
  final_distances_base(metric, e, n, sqrt_mag_a, dist);
}

double query_mag_a(const float *q, int dim) {
#pragma clang fp contract(off)
  double m = 0;
  for (int i = 0; i < dim; ++i) m = m + (double)q[i] * (double)q[i];
  return m;
}

// threshold + order + cut of ngh_graph_engine.dart:127,133-134 over candidate
// entries from any number of blocks.  mag_a: query_mag_a(query) when the caller has it already (the batched path
// computes it while it prepares the queries), else nullptr.
typedef std::pair<const BlockEntry *, uint32_t> EntryList;
int32_t finalize_query(int metric, int dim, const float *query, int32_t k, double thr, const EntryList *lists_p,
                       size_t n_lists, int64_t *out_ids, double *out_dist, const double *mag_a_known = nullptr) {
  struct {
    const EntryList *b, *e;
    const EntryList *begin() const { return b; }
    const EntryList *end() const { return e; }
  } lists{lists_p, lists_p + n_lists};
  const double mag_a = metric != TSH_METRIC_COSINE ? 0.0 : (mag_a_known ? *mag_a_known : query_mag_a(query, dim));
  const double sqrt_mag_a = std::sqrt(mag_a);
  static thread_local std::vector<Hit> hits;  // (one allocation per thread, not per query)
  static thread_local std::vector<double> dist;
  hits.clear();
  size_t total = 0;
  for (auto &l : lists) total += l.second;
  hits.reserve(total);
  const bool has_thr = !std::isnan(thr);
  for (auto &l : lists) {
    if (dist.size() < l.second) dist.resize(l.second);
    final_distances(metric, l.first, l.second, sqrt_mag_a, dist.data());
    for (uint32_t i = 0; i < l.second; ++i) {
      const double d = dist[i];
      if (has_thr && d > thr) continue;
      hits.push_back({dart_order_key(d), l.first[i].id});
    }
  }
  size_t r = std::min<size_t>(hits.size(), (size_t)std::max(k, 0));
  if (hits.size() <= 4 * r) {  // the usual case (k + a band's worth of candidates): one sort is cheapest
    sort_hits(hits.data(), hits.size());
  } else {
    std::nth_element(hits.begin(), hits.begin() + r, hits.end(), hit_less);
    std::sort(hits.begin(), hits.begin() + r, hit_less);
  }
  for (size_t i = 0; i < r; ++i) {
    out_ids[i] = hits[i].id;
    out_dist[i] = dart_order_key_to_double(hits[i].key);
  }
  for (size_t i = r; i < (size_t)std::max(k, 0); ++i) {  // unused slots read as "no row" (callers need not pre-fill)
    out_ids[i] = -1;
    out_dist[i] = std::nan("");
  }
  return (int32_t)r;
}
inline int32_t finalize_query(int metric, int dim, const float *query, int32_t k, double thr,
                              const std::vector<EntryList> &lists, int64_t *out_ids, double *out_dist) {
  return finalize_query(metric, dim, query, k, thr, lists.data(), lists.size(), out_ids, out_dist);
}

// ---- kernel dispatch ---------------------------------------------------------
inline int pick_nch(int d4) {
  int need = (d4 + 63) / 64;
  static const int opts[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16};
  for (int o : opts)
    if (o >= need) return o;
  return -1;
}

// the scan kernels' launchers live in their own translation unit (tsh_scan_tu.hip): 144 instantiations that
// compile beside this file instead of in front of it
// ---- per-search scratch: one context = one query in flight ---------------------
struct Ctx {
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t ev_done = nullptr;  // recorded after a job's last kernel
  hipEvent_t ev_scanned = nullptr;  // scan stream -> tail stream hand-off
  uint8_t *h_block_dev = nullptr;  // device-side address of h_block (zero-copy result stores)
  float *d_query = nullptr;  // ld floats
  float *h_query = nullptr;  // pinned, ld floats
  uint64_t *d_mask = nullptr;
  uint64_t *h_mask = nullptr;  // pinned
  int64_t mask_words = 0;
  uint64_t mask_epoch = 0;  // which caller mask the device copy holds
  uint32_t *d_list = nullptr, *h_list = nullptr;  // selective masks: the kept rows' ids (scan_list_kernel), h_ pinned
  uint32_t *h_list_dev = nullptr;                 // ... and mapped: the exact path's first scan of a list reads it in place
  int64_t list_cap = 0;
  uint64_t list_epoch = 0;
  uint32_t *d_keys = nullptr;
  uint32_t *d_gmin = nullptr;
  uint32_t *d_hist = nullptr;  // radix select of the fallback path (256 bins)
  int64_t tiles_cap = 0;
  uint8_t *d_block = nullptr;  // header + entries
  uint8_t *h_block = nullptr;  // pinned
  uint32_t *d_cand = nullptr;  // entries
  int64_t entries_cap = 0;
  // wide-band fallback
  uint32_t *d_big_rows = nullptr;
  BlockEntry *d_big_entries = nullptr;
  uint32_t *d_big_count = nullptr;
  int64_t big_cap = 0;
  BlockEntry *h_quar = nullptr, *h_quar_dev = nullptr;  // pinned + mapped: sums of the quarantined rows
  // short searches (tsh_exact.hip.h): per entry the order key of its exact distance and its two f64 sums
  uint64_t *d_xkey = nullptr;
  double *d_xsum = nullptr;
  uint64_t *d_xpick = nullptr;  // exact_pick_kernel: [0] its counter (zero between searches), [1 ..] E1's wave minima
  int64_t x_cap = 0;
  int64_t bytes = 0;
};

// Streams are a finite resource (CU-masked ones especially: about 85 open indexes with their own
// five streams plus context streams crashed the runtime), and everything on one GPU shares its HBM and
// CUs anyway -- so all shards of a device share ONE set, created on first use and never destroyed.
struct DeviceStreams {
  hipStream_t ingest = nullptr;
  hipStream_t scan = nullptr, scan2 = nullptr;  // pipeline streams (240-CU mask when cu_split)
  hipStream_t tail = nullptr, tail2 = nullptr;  // the other 16 CUs (two queues: the tails of consecutive queries overlap)
  hipStream_t batch = nullptr;                  // unmasked: matrix-core batches
  hipStream_t aux = nullptr;                    // rare synchronous work (fallback filter, bench hooks)
  hipStream_t upload = nullptr;                 // a batched call's queries on their way in, beside the call before it
  bool cu_split = false;
  std::mutex scan_mu;                           // one enqueue sequence (scan + hand-off + tail) at a time
  // small shards run their scans on two streams: which scans of either are still out (guarded by scan_mu).  A new
  // scan goes to the stream with fewer of them -- two scans side by side do not share the HBM evenly, and with strict
  // alternation one stream ended a call a scan or two behind the other, which sat idle meanwhile
  struct ScanOut {
    const void *owner;  // the context whose job this scan belongs to
    hipEvent_t ev;      // rides on the scan's dispatch packet
    int which;
  };
  std::vector<ScanOut> scan_out;
  int scans_out[2] = {0, 0};
  uint64_t enq_counter = 0;  // jobs enqueued on this device's streams so far (guarded by scan_mu)
  int rc = TSH_OK;
  std::string err;
  int users = 0;                                // shards alive on this device (device_streams / device_streams_release)
  int cus = 0;
};
DeviceStreams *device_streams(int device);  // defined after set_err / HIPCHK users below
void device_streams_release(int device);

struct Shard {
  int device = 0;
  int dim = 0, metric = 0, nch = 0;
  int64_t ld = 0;
  int64_t row_base = 0;  // global id of local row 0
  int64_t rows = 0;      // local rows (next local id)
  int64_t cap = 0;       // allocated rows (multiple of 64)
  float *d_rows = nullptr;
  float *d_inv_norm = nullptr;
  float *d_sqnorm = nullptr;  // |row|^2 (batched L2 key)
  uint64_t *d_live = nullptr;
  IngestStats *d_stats = nullptr;
  bool holds_streams = false;  // counted in its device's DeviceStreams::users
  uint32_t *d_tmp_u32 = nullptr;
  int64_t *d_del_ids = nullptr;  // tsh_index_set_deleted's id buffer, kept between calls
  int64_t del_ids_cap = 0;
  int64_t deleted = 0;
  bool all_live = true;  // every row in [0,rows) is present and not deleted
  float max_norm = 0.f, max_abs = 0.f;
  float min_norm = 0.f;  // smallest |row| seen (0 until rows exist, or when a zero row exists)
  uint32_t nonfinite_rows = 0, tiny_rows = 0;
  // Rows the f32 error model cannot cover (non-finite / > 1e15 elements, tiny cosine norms) are QUARANTINED: their
  // live bit stays clear, so no scan / batch kernel ever offers them, and every search adds their exact sums
  // (quarantine_kernel) to its candidates instead -- a few broken embeddings then do not push every search of
  // the shard onto the rerank-everything path.  Sorted local ids + device copy ([0] = count); changed under the
  // exclusive lock only.  Host-side searches hand the entries to the finaliser as an extra list; shard mode
  // (tsh_search_shard) appends them to the device blocks (quarantine_append_kernel).  More than
  // QUARANTINE_MAX of them fall back to safe mode.
  std::vector<uint32_t> quar_ids;
  uint32_t *d_quar = nullptr;
  uint32_t *d_irr = nullptr;  // 1 + QUARANTINE_MAX: count and ids of the irregular rows of one append
  hipStream_t ingest_stream = nullptr;  // all streams below belong to the device's DeviceStreams
  hipStream_t aux_stream = nullptr;
  // every query's scan -> select -> rerank runs on this one in-order stream,
  // back to back; only the small input / result copies use the context streams.
  // (Running a query's tail beside the next query's scan was measured 10-25x
  // slower per tail: each dependent load queues behind the scan's loads.)
  hipStream_t scan_stream = nullptr;
  // small shards alternate their scans between two streams (see job_enqueue)
  hipStream_t scan_stream2 = nullptr;
  uint64_t scan_seq = 0;  // guarded by scan_mu
  // When several queries are in flight, a query's select + rerank run here, on
  // CUs the scan stream's CU mask leaves free (2 per XCD), so they overlap the
  // next query's scan without queueing behind its loads on the same CU.
  hipStream_t tail_stream = nullptr, tail_stream2 = nullptr;
  uint64_t tail_seq = 0;  // guarded by scan_mu
  bool cu_split = false;
  hipStream_t batch_stream = nullptr;  // matrix-core batches: compute-bound, so all CUs (no mask)
  hipStream_t upload_stream = nullptr;  // H2D copies of a batched call's inputs (they overlap the call in front)
  std::mutex *scan_mu = nullptr;  // the device's (DeviceStreams): streams are shared by its shards
  DeviceStreams *dstreams = nullptr;
  std::atomic<int> inflight{0};

  RwLock mu;  // search: shared; append/delete: exclusive
  std::mutex ctx_mu;
  std::condition_variable ctx_cv;
  std::vector<std::unique_ptr<Ctx>> ctx_all;
  std::vector<Ctx *> ctx_free;
  std::atomic<uint64_t> mask_epoch_src{1};

  std::atomic<int64_t> c_searches{0}, c_scans{0}, c_batches{0}, c_fallbacks{0}, c_cands{0};
  std::atomic<int64_t> c_plane_fallbacks{0}, c_scan_fallbacks{0};  // batched calls degraded by a full device
  std::atomic<int64_t> c_list_scans{0};  // scans of a compacted row list (selective masks)
  std::atomic<int64_t> c_exact_scans{0};  // searches answered by the exact scan of a few thousand rows (tsh_exact.hip.h)
  int exact_rows = EX_MAX_ROWS;  // TSH_OPT_EXACT_SCAN_ROWS: searches that look at no more rows than this take that path
  bool exact_pick = true;        // TSH_OPT_EXACT_SELECT: the wide pick (exact_pick_kernel) behind the exact scan
  std::atomic<int64_t> c_pick_redone{0};  // picks whose cut bin overflowed the block: finished by exact_select_kernel
  int cus = 0;  // compute units of the shard's device (grid of the persistent key kernels)
  std::atomic<int> f16_strikes{0};      // batched calls in a row whose fp16 bands overflowed many candidate lists
  std::atomic<int> f16_denied_calls{0};  // auto key-kernel choice: bf16x3 instead of fp16 for this many more batched calls
  std::atomic<int> planes_denied{0};  // batched calls left that go straight to the f32 kernel (the copy did not fit)
  double scan_us_sum = 0;  // guarded by ctx_mu
  int64_t scan_us_samples = 0;
  int64_t bytes = 0;
  struct BatchCtx *batch = nullptr;  // matrix-core path scratch (created with the shard)
  // A second scratch set: two batched calls (two host threads, or one call per isolate) overlap -- the later one's
  // host preparation runs while the GPU works on the earlier one, whose finalisation runs while the GPU works on
  // the later one.  The GPU side stays one in-order stream: batch_enq_mu covers a call's enqueue sequence and the
  // upkeep of the converted planes.
  struct BatchCtx *batch2 = nullptr;
  std::mutex batch_enq_mu;
  // bf16 (hi, lo) planes of the rows for the bf16x3 batch kernel; built lazily by the first
  // batch search, kept current from split_valid (guarded by batch_enq_mu under a shared s->mu;
  // appends lower split_valid under the exclusive lock)
  u32x4 *d_split = nullptr;
  int64_t split_cap = 0;    // rows allocated
  int64_t split_valid = 0;  // rows [0, split_valid) are converted
  int batch_kernel = 3;     // TSH_OPT_BATCH_KERNEL: 0 f32 MFMA, 1 bf16x3, 2 f16, 3 auto (cosine: f16, else bf16x3)
  // TSH_OPT_BATCH_HUB: the hub rows' bound beside the sample's (fp16 keys, L2 / inner product).  OFF by default: measured
  // on the bench's L2 corpus (norms U(0.5, 2), 1 M x 768, 1024 queries; same box, alternating, tools/r6_hub_ab.sh) it cuts
  // the key passes 1510-1536 -> 1463-1468 us -- the 4096 shortest rows reach |v| <= 0.506, the bound lets ~300 rows per query
  // through where the estimate lets 480 -- and the call as a whole gains nothing (520-530 k against 529-533 k queries/s):
  // sixteen more dense tiles and a second order statistic per query cost what the thinner epilogue saves; inner product
  // and unit-norm rows: no difference; norms U(0.1, 3.2): + 1.8 %.
  bool batch_hub = false;
  std::atomic<int> batch_kernel_last{-1};  // variant the last batched search ran
  int split_mode = 0;       // which kernel the planes were built for (1 / 2); 0 = none
  int64_t split_bytes = 0;
  int split_exp = 0;        // f16 planes: rows were scaled by 2^split_exp
  // TSH_OPT_BATCH_GROUP (default on): the fp16 plane of an L2 / inner-product shard holds its rows by norm inside
  // blocks of PG_ROWS (plane_group_kernel, tsh_batch_f16.hip.h): d_perm[position] = row, d_psq[position] = that row's
  // |v|^2.  Kept current with the plane (same guards); split_grouped = how the plane as built is ordered.
  bool batch_group = true;
  bool split_grouped = false;
  uint32_t *d_perm = nullptr;
  float *d_psq = nullptr;
  int64_t perm_cap = 0, perm_bytes = 0;  // positions allocated
  // The HUB rows of an L2 / inner-product shard (round 6): the few thousand rows whose norm alone puts them near every
  // query -- the shortest (L2) resp. the longest (inner product) --, as a gathered fp16 copy beside the planes.  A batched
  // call scores them densely beside its sample; their k-th smallest key is a PROVEN bound on the k-th key overall
  // (SampleSelArgs::hub_dense).  Built by the first batched call that wants it (guarded by batch_enq_mu like the planes);
  // rows appended later are simply not in it (any subset of the rows gives a valid bound) until they are a quarter of
  // the shard; an overwrite of rows it may hold drops it.
  u32x4 *d_hub = nullptr;
  uint32_t *d_hub_ids = nullptr;  // plane position -> local row id
  float *d_hub_sq = nullptr;      // |v|^2 of the hub rows, in plane order
  int64_t hub_rows_built = -1;    // rows the shard had when it was built (-1: none)
  int hub_n = 0, hub_exp = 0, hub_chunks = 0;
  int64_t hub_bytes = 0;

  bool safe_mode() const {
    if (nonfinite_rows) return true;
    if (max_abs > BIG_ABS) return true;
    if (metric == TSH_METRIC_COSINE && tiny_rows) return true;
    return false;
  }
};

// The stream sets live as long as the process -- except the CU-masked streams: left to the runtime's own teardown
// they crash a python host at exit under rocprofv3 (tools/exitcheck.sh: SIGSEGV in __cxa_finalize after the tool
// wrote its output; plain streams are fine).  They are destroyed when the LAST shard on their device is (and made
// again for the next one), so a host that destroys its handles -- every host should -- has none left at exit.  The
// atexit handler only covers leaked handles; it is registered after the first HIP call, so it runs before the
// runtime's exit handlers.  (Round 3: with torch in the process rocprofv3 finalises its tool BEFORE that handler, and
// a HIP call after that aborts -- tools/exitcheck_bench.sh; hence the reference count, which leaves the handler idle.)
static std::mutex g_streams_mu;
static std::map<int, DeviceStreams *> *g_stream_sets = nullptr;  // never freed
static void destroy_masked_streams(DeviceStreams *ds) {  // g_streams_mu held, the device current
  for (hipStream_t *st : {&ds->scan, &ds->scan2, &ds->tail, &ds->tail2})
    if (*st) {
      (void)hipStreamSynchronize(*st);
      (void)hipStreamDestroy(*st);
      *st = nullptr;
    }
  ds->cu_split = false;
}
static void destroy_masked_streams_at_exit() {
  std::lock_guard<std::mutex> lk(g_streams_mu);
  if (!g_stream_sets) return;
  for (auto &kv : *g_stream_sets) {
    DeviceStreams *ds = kv.second;
    if (!ds->cu_split || hipSetDevice(kv.first) != hipSuccess) continue;
    destroy_masked_streams(ds);
  }
}

// the pipeline / tail streams: CU-masked where the device allows it, one plain stream otherwise
static hipError_t make_scan_streams(DeviceStreams *ds) {
  // CU mask bit i = CU slot i/8 of XCD i%8 on MI355X (measured, tools/cumask_probe.hip):
  // bits 0..15 = two CUs of every XCD, reserved for the tails.
  const char *env = probe_env("TSH_NO_CU_SPLIT");
  const int cus = ds->cus;
  if (!(env && env[0] == '1') && cus >= 64 && cus % 32 == 0) {
    std::vector<uint32_t> scan_mask((size_t)cus / 32, 0xFFFFFFFFu), tail_mask((size_t)cus / 32, 0u);
    scan_mask[0] = 0xFFFF0000u;
    tail_mask[0] = 0x0000FFFFu;
    if (hipExtStreamCreateWithCUMask(&ds->scan, (uint32_t)scan_mask.size(), scan_mask.data()) == hipSuccess &&
        hipExtStreamCreateWithCUMask(&ds->scan2, (uint32_t)scan_mask.size(), scan_mask.data()) == hipSuccess &&
        hipExtStreamCreateWithCUMask(&ds->tail, (uint32_t)tail_mask.size(), tail_mask.data()) == hipSuccess &&
        hipExtStreamCreateWithCUMask(&ds->tail2, (uint32_t)tail_mask.size(), tail_mask.data()) == hipSuccess) {
      ds->cu_split = true;
      return hipSuccess;
    }
    if (ds->scan) (void)hipStreamDestroy(ds->scan);
    if (ds->scan2) (void)hipStreamDestroy(ds->scan2);
    if (ds->tail) (void)hipStreamDestroy(ds->tail);
    if (ds->tail2) (void)hipStreamDestroy(ds->tail2);
    ds->scan = ds->scan2 = ds->tail = ds->tail2 = nullptr;
    (void)hipGetLastError();
  }
  return hipStreamCreateWithFlags(&ds->scan, hipStreamNonBlocking);
}

// One more shard on `device`: its stream set (made on first use; the masked streams made again if the last shard
// before this one took them along).  Pair with device_streams_release.
DeviceStreams *device_streams(int device) {
  std::lock_guard<std::mutex> lk(g_streams_mu);
  if (!g_stream_sets) {
    g_stream_sets = new std::map<int, DeviceStreams *>();
    atexit(destroy_masked_streams_at_exit);
  }
  std::map<int, DeviceStreams *> &sets = *g_stream_sets;
  auto it = sets.find(device);
  if (it != sets.end()) {
    DeviceStreams *ds = it->second;
    if (ds->rc == TSH_OK && !ds->scan) {
      hipError_t e = hipSetDevice(device);
      if (e == hipSuccess) e = make_scan_streams(ds);
      if (e != hipSuccess) {
        ds->rc = e == hipErrorOutOfMemory ? TSH_E_OOM : TSH_E_HIP;
        ds->err = std::string("hipStreamCreate failed: ") + hipGetErrorString(e);
      }
    }
    if (ds->rc == TSH_OK) ++ds->users;
    return ds;
  }
  DeviceStreams *ds = new DeviceStreams();
  sets[device] = ds;
  auto fail = [&](hipError_t e, const char *what) {
    ds->rc = e == hipErrorOutOfMemory ? TSH_E_OOM : TSH_E_HIP;
    ds->err = std::string(what) + " failed: " + hipGetErrorString(e);
    return ds;
  };
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return fail(e, "hipSetDevice");
  if ((e = hipStreamCreateWithFlags(&ds->ingest, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return fail(e, "hipGetDeviceProperties");
  ds->cus = prop.multiProcessorCount;
  if ((e = make_scan_streams(ds)) != hipSuccess) return fail(e, "hipStreamCreate");
  if ((e = hipStreamCreateWithFlags(&ds->batch, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
  if ((e = hipStreamCreateWithFlags(&ds->aux, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
  if ((e = hipStreamCreateWithFlags(&ds->upload, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
  ++ds->users;
  return ds;
}

void device_streams_release(int device) {
  std::lock_guard<std::mutex> lk(g_streams_mu);
  if (!g_stream_sets) return;
  auto it = g_stream_sets->find(device);
  if (it == g_stream_sets->end()) return;
  DeviceStreams *ds = it->second;
  if (ds->users > 0 && --ds->users == 0 && ds->cu_split && hipSetDevice(device) == hipSuccess) destroy_masked_streams(ds);
}

int shard_init(Shard *s) {
  HIPCHK(hipSetDevice(s->device));
  DeviceStreams *ds = device_streams(s->device);
  if (ds->rc != TSH_OK) return set_err(ds->rc, "%s", ds->err.c_str());
  s->ingest_stream = ds->ingest;
  s->scan_stream = ds->scan;
  s->scan_stream2 = ds->scan2;
  s->tail_stream = ds->tail;
  s->tail_stream2 = ds->tail2;
  s->batch_stream = ds->batch;
  s->aux_stream = ds->aux;
  s->upload_stream = ds->upload;
  s->cu_split = ds->cu_split;
  s->cus = ds->cus;
  s->scan_mu = &ds->scan_mu;
  s->dstreams = ds;
  s->holds_streams = true;
  HIPCHK(hipMalloc(&s->d_stats, sizeof(IngestStats)));
  HIPCHK(hipMemset(s->d_stats, 0, sizeof(IngestStats)));
  HIPCHK(hipMalloc(&s->d_tmp_u32, 64));
  HIPCHK(hipMalloc(&s->d_quar, (1 + QUARANTINE_MAX) * sizeof(uint32_t)));
  HIPCHK(hipMalloc(&s->d_irr, (1 + QUARANTINE_MAX) * sizeof(uint32_t)));
  HIPCHK(hipMemset(s->d_quar, 0, 4));
  return TSH_OK;
}

int shard_reserve(Shard *s, int64_t want_rows) {
  if (want_rows <= s->cap) return TSH_OK;
  int64_t ncap = std::max<int64_t>(round_up(want_rows, 64), 64);
  if (s->cap > 0) ncap = std::max<int64_t>(ncap, round_up(s->cap + s->cap / 2, 64));
  HIPCHK(hipSetDevice(s->device));
  float *nrows = nullptr, *ninv = nullptr, *nsq = nullptr;
  uint64_t *nlive = nullptr;
  size_t row_bytes = (size_t)ncap * (size_t)s->ld * sizeof(float);
  HIPCHK(hipMalloc(&nrows, row_bytes));
  if (hipMalloc(&ninv, (size_t)ncap * sizeof(float)) != hipSuccess ||
      hipMalloc(&nsq, (size_t)ncap * sizeof(float)) != hipSuccess ||
      hipMalloc(&nlive, (size_t)(ncap / 64) * sizeof(uint64_t)) != hipSuccess) {
    hipFree(nrows);
    if (ninv) hipFree(ninv);
    if (nsq) hipFree(nsq);
    return set_err(TSH_E_OOM, "hipMalloc failed for %lld rows", (long long)ncap);
  }
  hipStream_t st = s->ingest_stream;
  size_t old_bytes = (size_t)s->cap * (size_t)s->ld * sizeof(float);
  if (s->cap > 0) {
    HIPCHK(hipMemcpyAsync(nrows, s->d_rows, old_bytes, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(ninv, s->d_inv_norm, (size_t)s->cap * sizeof(float),
                          hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(nsq, s->d_sqnorm, (size_t)s->cap * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(nlive, s->d_live, (size_t)(s->cap / 64) * sizeof(uint64_t),
                          hipMemcpyDeviceToDevice, st));
  }
  // rows past the old capacity read as zeros until written (tiles are whole)
  HIPCHK(hipMemsetAsync((char *)nrows + old_bytes, 0, row_bytes - old_bytes, st));
  HIPCHK(hipMemsetAsync(ninv + s->cap, 0, (size_t)(ncap - s->cap) * sizeof(float), st));
  HIPCHK(hipMemsetAsync(nsq + s->cap, 0, (size_t)(ncap - s->cap) * sizeof(float), st));
  HIPCHK(hipMemsetAsync(nlive + s->cap / 64, 0, (size_t)((ncap - s->cap) / 64) * sizeof(uint64_t), st));
  HIPCHK(hipStreamSynchronize(st));
  if (s->d_rows) hipFree(s->d_rows);
  if (s->d_inv_norm) hipFree(s->d_inv_norm);
  if (s->d_sqnorm) hipFree(s->d_sqnorm);
  if (s->d_live) hipFree(s->d_live);
  s->d_rows = nrows;
  s->d_inv_norm = ninv;
  s->d_sqnorm = nsq;
  s->d_live = nlive;
  s->bytes += (int64_t)(row_bytes - old_bytes) + (ncap - s->cap) * 8 + (ncap - s->cap) / 8;
  s->cap = ncap;
  return TSH_OK;
}

// rows [first, first+n) local ids; src host or device
// device copy of the quarantine list; caller holds s->mu exclusively
int quarantine_upload(Shard *s) {
  std::vector<uint32_t> h(1 + s->quar_ids.size());
  h[0] = (uint32_t)s->quar_ids.size();
  std::copy(s->quar_ids.begin(), s->quar_ids.end(), h.begin() + 1);
  HIPCHK(hipMemcpyAsync(s->d_quar, h.data(), h.size() * 4, hipMemcpyHostToDevice, s->ingest_stream));
  HIPCHK(hipStreamSynchronize(s->ingest_stream));
  return TSH_OK;
}

// indices into quar_ids of the quarantined rows a caller's mask lets through (mask_words: the shard's slice,
// bit r = local row r; NULL = all)
void quarantine_select(const Shard *s, const uint64_t *mask_words, std::vector<uint32_t> *sel) {
  sel->clear();
  for (uint32_t i = 0; i < (uint32_t)s->quar_ids.size(); ++i) {
    const uint32_t r = s->quar_ids[i];
    if (!mask_words || ((mask_words[r >> 6] >> (r & 63)) & 1ull)) sel->push_back(i);
  }
}

int shard_append(Shard *s, int64_t first, int64_t n, const float *src, bool src_is_device) {
  if (n <= 0) return TSH_OK;
  int rc = shard_reserve(s, first + n);
  if (rc) return rc;
  HIPCHK(hipSetDevice(s->device));
  hipStream_t st = s->ingest_stream;
  float *dst = s->d_rows + first * s->ld;
  hipMemcpyKind kind = src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  if (s->ld == s->dim) {
    HIPCHK(hipMemcpyAsync(dst, src, (size_t)n * s->dim * sizeof(float), kind, st));
  } else {
    HIPCHK(hipMemcpy2DAsync(dst, (size_t)s->ld * sizeof(float), src, (size_t)s->dim * sizeof(float),
                            (size_t)s->dim * sizeof(float), (size_t)n, kind, st));
  }
  // quarantine: rows of [first, first + n) are replaced, so their old entries go; the kernel lists the new ones
  uint32_t irr_cap = 0;
  bool quar_dirty = false;
  {
    auto lo = std::lower_bound(s->quar_ids.begin(), s->quar_ids.end(), (uint32_t)first);
    auto hi = std::lower_bound(lo, s->quar_ids.end(), (uint32_t)std::min<int64_t>(first + n, 0xFFFFFFFFll));
    if (lo != hi) {
      s->quar_ids.erase(lo, hi);
      quar_dirty = true;
    }
    irr_cap = QUARANTINE_MAX - (uint32_t)s->quar_ids.size();
    HIPCHK(hipMemsetAsync(s->d_irr, 0, 4, st));
  }
  int blocks = (int)std::min<int64_t>((n + 3) / 4, 2048);
  ingest_kernel<<<blocks, 256, 0, st>>>(s->d_rows, s->ld, s->dim, first, n, s->d_inv_norm, s->d_sqnorm, s->d_stats,
                                        s->d_irr, irr_cap,
                                        s->metric == TSH_METRIC_COSINE ? 1 : 0);
  int lb = (int)std::min<int64_t>(((first + n - 1) / 64 - first / 64 + 1 + 255) / 256, 1024);
  live_range_kernel<<<lb, 256, 0, st>>>(s->d_live, first, n, 1);
  IngestStats hs;
  uint32_t n_irr = 0;
  HIPCHK(hipMemcpyAsync(&hs, s->d_stats, sizeof hs, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&n_irr, s->d_irr, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  n_irr = std::min(n_irr, irr_cap);
  if (n_irr) {
    std::vector<uint32_t> ids(n_irr);
    live_clear_u32_kernel<<<(n_irr + 255) / 256, 256, 0, st>>>(s->d_live, s->d_irr + 1, n_irr);
    HIPCHK(hipMemcpyAsync(ids.data(), s->d_irr + 1, (size_t)n_irr * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    s->quar_ids.insert(s->quar_ids.end(), ids.begin(), ids.end());
    std::sort(s->quar_ids.begin(), s->quar_ids.end());
    s->all_live = false;
    quar_dirty = true;
  }
  if (quar_dirty) {
    rc = quarantine_upload(s);
    if (rc) return rc;
  }
  uint32_t mb = hs.max_norm_bits, ab = hs.max_abs_bits;
  memcpy(&s->max_norm, &mb, 4);
  memcpy(&s->max_abs, &ab, 4);
  s->nonfinite_rows = hs.nonfinite_rows;
  s->tiny_rows = hs.tiny_rows;
  {
    uint32_t nb = hs.inv_min_norm_bits ? ~hs.inv_min_norm_bits : 0u;
    memcpy(&s->min_norm, &nb, 4);
  }
  if (first > s->rows) s->all_live = false;  // gap of absent rows
  if (first + n > s->rows) s->rows = first + n;
  s->split_valid = std::min(s->split_valid, first);  // overwritten / new rows need re-splitting
  if (first < s->hub_rows_built) s->hub_rows_built = -1;  // (an overwritten row may be a hub row: its copy is stale)
  return TSH_OK;
}

void ctx_free_all(Ctx *c) {
  if (c->ev0) hipEventDestroy(c->ev0);
  if (c->ev1) hipEventDestroy(c->ev1);
  if (c->ev_done) hipEventDestroy(c->ev_done);
  if (c->ev_scanned) hipEventDestroy(c->ev_scanned);
  hipFree(c->d_query);
  hipHostFree(c->h_query);
  hipFree(c->d_mask);
  hipHostFree(c->h_mask);
  hipFree(c->d_list);
  hipHostFree(c->h_list);
  hipFree(c->d_keys);
  hipFree(c->d_gmin);
  hipFree(c->d_hist);
  hipFree(c->d_block);
  hipHostFree(c->h_block);
  hipHostFree(c->h_quar);
  hipFree(c->d_cand);
  hipFree(c->d_big_rows);
  hipFree(c->d_big_entries);
  hipFree(c->d_big_count);
  hipFree(c->d_xkey);
  hipFree(c->d_xsum);
  hipFree(c->d_xpick);
}

int ctx_prepare(Shard *s, Ctx *c, int32_t entries, bool need_mask) {
  HIPCHK(hipSetDevice(s->device));
  if (!c->ev0) {
    HIPCHK(hipEventCreate(&c->ev0));
    HIPCHK(hipEventCreate(&c->ev1));
    HIPCHK(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming | (blocking_wait() ? hipEventBlockingSync : 0)));
    HIPCHK(hipEventCreateWithFlags(&c->ev_scanned, hipEventDisableTiming));
    HIPCHK(hipMalloc(&c->d_query, (size_t)s->ld * sizeof(float)));
    HIPCHK(hipHostMalloc(&c->h_query, (size_t)s->ld * sizeof(float), hipHostMallocDefault));
    HIPCHK(hipMalloc(&c->d_big_count, 64));
    c->bytes += s->ld * 4;
  }
  int64_t tiles = s->cap / 64;
  if (tiles > c->tiles_cap) {
    hipFree(c->d_keys);
    hipFree(c->d_gmin);
    c->d_keys = nullptr;
    c->d_gmin = nullptr;
    HIPCHK(hipMalloc(&c->d_keys, (size_t)tiles * 64 * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&c->d_gmin, (size_t)tiles * sizeof(uint32_t)));
    c->bytes += (tiles - c->tiles_cap) * 65 * 4;
    c->tiles_cap = tiles;
  }
  if (need_mask && tiles > c->mask_words) {
    hipFree(c->d_mask);
    hipHostFree(c->h_mask);
    c->d_mask = nullptr;
    c->h_mask = nullptr;
    HIPCHK(hipMalloc(&c->d_mask, (size_t)tiles * 8));
    HIPCHK(hipHostMalloc(&c->h_mask, (size_t)tiles * 8, hipHostMallocDefault));
    c->bytes += (tiles - c->mask_words) * 8;
    c->mask_words = tiles;
    c->mask_epoch = 0;
  }
  if (entries > c->entries_cap) {
    hipFree(c->d_block);
    hipHostFree(c->h_block);
    hipFree(c->d_cand);
    c->d_block = nullptr;
    c->h_block = nullptr;
    c->d_cand = nullptr;
    size_t bb = (size_t)tsh_candidate_block_bytes(entries);
    HIPCHK(hipMalloc(&c->d_block, bb));
    HIPCHK(hipHostMalloc(&c->h_block, bb, hipHostMallocMapped));
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void **>(&c->h_block_dev), c->h_block, 0));
    HIPCHK(hipMalloc(&c->d_cand, (size_t)entries * sizeof(uint32_t)));
    c->bytes += (int64_t)(bb + 4 * (size_t)entries);
    c->entries_cap = entries;
  }
  return TSH_OK;
}

Ctx *ctx_acquire(Shard *s, bool block) {
  std::unique_lock<std::mutex> lk(s->ctx_mu);
  for (;;) {
    if (!s->ctx_free.empty()) {
      Ctx *c = s->ctx_free.back();
      s->ctx_free.pop_back();
      return c;
    }
    if ((int)s->ctx_all.size() < MAX_CTX) {
      s->ctx_all.emplace_back(new Ctx());
      return s->ctx_all.back().get();
    }
    if (!block) return nullptr;
    s->ctx_cv.wait(lk);
  }
}
void ctx_release(Shard *s, Ctx *c) {
  if (s->dstreams) {  // whatever the device still books under this context's events is over (or given up)
    std::lock_guard<std::mutex> lk(*s->scan_mu);
    std::vector<DeviceStreams::ScanOut> &so = s->dstreams->scan_out;
    for (size_t i = 0; i < so.size();)
      if (so[i].owner == c) {
        s->dstreams->scans_out[so[i].which]--;
        so[i] = so.back();
        so.pop_back();
      } else {
        ++i;
      }
  }
  {
    std::lock_guard<std::mutex> lk(s->ctx_mu);
    s->ctx_free.push_back(c);
  }
  s->ctx_cv.notify_one();
}

// error band of the f32 ranking key (DESIGN.md "error model")
struct Band {
  float eps_rel = 0.f, delta_abs = 0.f;
  int force_all = 0;
};
Band compute_band(const Shard *s, const float *q) {
  Band b;
  if (s->safe_mode()) {
    b.force_all = 1;
    return b;
  }
  double qn2 = 0;
  for (int i = 0; i < s->dim; ++i) {
    float a = std::fabs(q[i]);
    if (!(a <= BIG_ABS)) {
      b.force_all = 1;  // inf / nan / huge query element
      return b;
    }
    qn2 += (double)q[i] * (double)q[i];
  }
  double qn = std::sqrt(qn2) * (1.0 + 1e-6);
  const double u2 = 1.1920928955078125e-07;  // 2^-23 = 2 * unit roundoff (safety factor 2)
  if (s->metric == TSH_METRIC_L2) {
    double eps = (4.0 * s->nch + 8.0) * u2;
    b.eps_rel = (float)(3.0 * eps);
    b.delta_abs = (float)((double)s->dim * 7.9e-31);  // d * 2^-100: underflow slack
  } else if (s->metric == TSH_METRIC_IP) {
    double gam = (4.0 * s->nch + 6.0) * u2;
    double delta = gam * qn * (double)s->max_norm * (1.0 + 1e-6) + (double)s->dim * 7.5e-37;
    b.delta_abs = (float)(2.0 * delta * 1.0001);
    if (!(b.delta_abs < 3.0e38f)) b.force_all = 1;
  } else {
    double gam = (4.0 * s->nch + 6.0) * u2;
    double delta = qn * (gam + 4.76837158203125e-07 /*2^-21*/) + (double)s->dim * 7.5e-37;
    b.delta_abs = (float)(2.0 * delta * 1.0001);
  }
  return b;
}

void fill_scan_args(const Shard *s, const Ctx *c, bool masked, bool user_mask, ScanArgsQ *aq) {
  ScanArgs *a = &aq->a;
  a->rows = s->d_rows;
  a->query = c->d_query;
  a->query_out = nullptr;
  a->inv_norm = s->d_inv_norm;
  a->live = s->d_live;
  a->mask = (masked && user_mask) ? c->d_mask : nullptr;
  a->keys = c->d_keys;
  a->gmin = c->d_gmin;
  a->ld = s->ld;
  a->n = s->rows;
  a->d4 = (int32_t)(s->ld / 4);
  a->n_tiles = (int32_t)((s->rows + 63) / 64);
  a->list = nullptr;
}

// slice the caller's GLOBAL keep mask into this shard's tile words
void slice_mask(const Shard *s, const uint8_t *mask, uint64_t *out_words, int64_t n_words) {
  int64_t rows = s->rows;
  memset(out_words, 0, (size_t)n_words * 8);
  uint8_t *ob = reinterpret_cast<uint8_t *>(out_words);
  int64_t nbytes = (rows + 7) / 8;
  if ((s->row_base & 7) == 0) {
    memcpy(ob, mask + s->row_base / 8, (size_t)nbytes);
  } else {
    int sh = (int)(s->row_base & 7);
    const uint8_t *src = mask + s->row_base / 8;
    int64_t src_last = (s->row_base + rows - 1) / 8 - s->row_base / 8;  // last valid src byte index
    for (int64_t i = 0; i < nbytes; ++i) {
      unsigned lo = src[i] >> sh;
      unsigned hi = (i + 1 <= src_last) ? (unsigned)(src[i + 1] << (8 - sh)) : 0u;
      ob[i] = (uint8_t)(lo | hi);
    }
  }
}

// One query in flight on one context.  Its three kernels (scan, select,
// rerank) are enqueued back to back on the shard's single in-order pipeline
// stream with NO copy and NO cross-stream dependency between them: the query
// rides in the scan kernel's argument segment, the result block is stored
// straight into pinned host memory, completion is one event.
struct Job {
  Ctx *c = nullptr;
  int32_t k = 0, entries = 0;
  bool masked = false, user_mask = false;
  uint8_t *dev_target = nullptr;  // shard mode: caller's device block (header + entries land there)
  bool timed = false;             // ev0/ev1 bracket this job's scan kernel
  bool counted = false;           // contributes to Shard::inflight
  float eps_rel = 0.f, delta_abs = 0.f;  // this query's error band (for the fallback's own threshold)
  bool force_all = false;
  int32_t list_tiles = 0;  // > 0: a list scan -- the context's keys / gmin are in list order, that many tiles of them
  const uint64_t *d_mask = nullptr;  // the caller's mask words on the device: the context's copy, or a mask handle's
  const uint32_t *d_list = nullptr;  // the scanned list on the device: the context's copy, or a mask handle's
  bool exact = false;      // answered by exact_scan_kernel + exact_select_kernel: the block is final, no f32 keys exist
  bool picked = false;     // ... by exact_pick_kernel instead of exact_select_kernel: a cut bin too full for the block is
  ExactSelArgs xsel{};     // finished by exact_select_kernel on the same keys (these arguments), job_finish
  bool leave_overflow = false;  // shard mode under TSH_OPT_EXCHANGE_AHEAD: an exchange enqueued behind this job's kernels
                                // may be reading the device block when the host looks at it -- a block whose list
                                // overflowed is then NOT rewritten by the wide-band pass (a peer could gather a new
                                // header over old entries); it keeps FLAG_LIST_OVERFLOW, which every rank answers by
                                // redoing the group with larger blocks, not ahead
  hipStream_t last_stream = nullptr;  // where the job's last kernel was enqueued (ev_done rides on it)
  uint64_t enq_seq = 0;               // ... and its place in the device's enqueue order (DeviceStreams::enq_counter)
  std::vector<uint32_t> quar_sel;  // entries of c->h_quar that belong to this query's candidates
};

// A selective caller mask as a list: local ids of the kept rows, ascending, padded with 0xFFFFFFFF to whole tiles of
// 64.  Made once per call (shard_search_blocks) and shared by the call's queries; nullptr = scan by tiles.
struct RowList {
  const uint32_t *ids = nullptr;    // on the host: every context of the call uploads it once ...
  const uint32_t *d_ids = nullptr;  // ... or resident on the device (a mask handle's, compacted there): read in place
  int32_t padded = 0;  // entries incl. padding (multiple of 64)
};

// ---- mask handles (tsh_mask_create, include/tostore_hip.h): one shard's part ------------------------------------
// The caller's bitmap sliced to this shard's rows, as the kernels read it (word t bit r = local row 64 t + r), on
// the host (the quarantined rows are matched against it there; the batched path places its sample window by it) and
// on the device (dense masked scans, the batched epilogue); and, when the mask is selective, the ascending list of
// its kept rows -- compacted on the device (tsh_mask.hip.h), never on the host.  Built for the rows the shard had at
// the time; a search that finds the shard grown rebuilds it first (mask_part).
struct MaskPart {
  int device = 0;
  std::atomic<int64_t> built_rows{-1};  // rows of the shard the part was built for (-1: not built)
  int32_t n_tiles = 0;
  int64_t kept = 0;                // set bits (tombstones not subtracted)
  std::vector<uint64_t> h_words;   // n_tiles words
  std::vector<int32_t> pre;        // pre[t] = kept rows in words [0, t): n_tiles + 1 entries
  uint64_t *d_words = nullptr;
  int64_t words_cap = 0;
  uint32_t *d_list = nullptr;      // kept rows' local ids, ascending, padded with 0xFFFFFFFF to a multiple of 64
  int64_t list_cap = 0;
  int32_t list_padded = 0;         // 0: no list (the mask keeps too many rows for one to pay, or none)
  uint32_t *d_bsum = nullptr;      // M1's per-workgroup counts + one word for M2's total
  int64_t bsum_cap = 0;
  int64_t bytes = 0;               // device bytes held
};
// the mask of a search, as the shard-level functions take it
struct MaskSrc {
  const uint8_t *bytes = nullptr;  // the caller's GLOBAL bitmap (pointer form: sliced, counted, listed per call) ...
  const MaskPart *part = nullptr;  // ... or this shard's part of a mask handle (resident)
  MaskSrc() {}
  MaskSrc(const uint8_t *b) : bytes(b) {}  // NOLINT: the pointer form converts where a mask is passed on
  explicit MaskSrc(const MaskPart *p) : part(p) {}
  explicit operator bool() const { return bytes != nullptr || part != nullptr; }
};

void launch_select(const SelectArgs &se, int32_t n_tiles, hipStream_t st) {
  // a few hundred tiles (an index of some ten thousand rows): four waves instead of sixteen synchronise
  // faster (10 k x 128: p50 42.8 -> 39.8 us); from a few thousand tiles on the wide workgroup wins
  // (fewer tiles than k: the tile-minimum bound says nothing, every tile is a hit and step (f) bisects over all of
  // them -- from registers only while they fit a workgroup's: 64 tiles for four waves, 256 for sixteen.  A list scan
  // of a few thousand kept rows lands exactly there: 79 tiles, k = 100 took 85 us per query instead of 30)
  const bool all_hit = n_tiles < se.k && n_tiles > 64;
  // (a list scan's tile minima bound the k-th key loosely -- its tiles hold the kept rows in id order, every one of them
  // a mix of near and far rows: a 1 % mask of 1 M rows is 157 tiles of which ~100 are hits, and step (f) bisects over
  // them from registers only with sixteen waves)
  const bool list_many_hits = se.list != nullptr && n_tiles > 64;
  if (n_tiles <= 512 && se.k <= 256 && !all_hit && !list_many_hits) select_kernel<256, true><<<1, 256, 0, st>>>(se);
  else if (n_tiles <= SEL_VPT * SEL_THREADS) select_kernel<SEL_THREADS, true><<<1, SEL_THREADS, 0, st>>>(se);
  else select_kernel<SEL_THREADS, false><<<1, SEL_THREADS, 0, st>>>(se);
}

// shard mode: the quarantined rows go into the job's device block
void launch_quarantine_append(Shard *s, Ctx *c, Job *j, hipStream_t st) {
  QuarAppendArgs qa{};
  qa.rows = s->d_rows;
  qa.Q = c->d_query;
  qa.list = s->d_quar;
  qa.mask = j->user_mask ? j->d_mask : nullptr;
  qa.blocks = j->dev_target;
  qa.ld = s->ld;
  qa.ldq = s->ld;
  qa.row_base = s->row_base;
  qa.block_bytes = 0;
  qa.dim = s->dim;
  qa.entries = j->entries;
  qa.metric = s->metric;
  quarantine_append_kernel<<<dim3((unsigned)((s->quar_ids.size() + 63) / 64), 1), 64, 0, st>>>(qa);
}

// mask_words: this shard's slice of the caller mask (host), or NULL; epoch
// identifies it so a context uploads it once per call
// rows_est: rows the scan will actually read (popcount of the caller's mask; <= 0: all of them)
int ctx_reserve_list(Ctx *c, int64_t padded) {
  if (padded <= c->list_cap) return TSH_OK;
  hipFree(c->d_list);
  hipHostFree(c->h_list);
  c->d_list = c->h_list = c->h_list_dev = nullptr;
  c->bytes -= c->list_cap * 4;
  c->list_cap = 0;
  c->list_epoch = 0;
  const int64_t want = round_up(padded + padded / 2, 4096);
  HIPCHK(hipMalloc(&c->d_list, (size_t)want * sizeof(uint32_t)));
  HIPCHK(hipHostMalloc(&c->h_list, (size_t)want * sizeof(uint32_t), hipHostMallocMapped));
  HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void **>(&c->h_list_dev), c->h_list, 0));
  c->list_cap = want;
  c->bytes += want * 4;
  return TSH_OK;
}

// (list_mask_bits / popcount_words: tsh_host_sync.h -- pure host functions, tested on the CPU)
// A selective mask (a WHERE clause that keeps a few percent of the rows) is scanned as a LIST of row ids: scan_list_kernel
// gathers the kept rows, eight per wave, instead of walking tiles that are mostly dead (tsh_kernels.hip.h).  The list
// pays below one kept row in list_div (tools/r4_list_probe.sh); TSH_LIST_DIV=0 switches it off.  -> ids filled (padded
// with 0xFFFFFFFF to whole tiles) and true when the scan should use them.
bool exact_applies(const Shard *s, int64_t n_exam, int32_t k, int32_t entries);
inline int64_t mask_list_div() {
  static const int64_t v = probe_env("TSH_LIST_DIV") ? atoll(probe_env("TSH_LIST_DIV")) : 24;
  return v;
}
// does a search with this many kept rows scan them as a list?  (the rule of both forms of a mask: pointer and handle)
bool row_list_pays(const Shard *s, int64_t rows_kept, int32_t k, int32_t entries) {
  const int64_t list_div = mask_list_div();
  if (list_div <= 0 || s->rows < 4096 || rows_kept <= 0) return false;
  // few enough kept rows for the exact path (tsh_exact.hip.h): their list is all that search reads, at any row width
  // and any selectivity (a shard small enough for that path whole needs no list: it tests the mask row by row)
  const bool by_exact = !exact_applies(s, s->rows, k, entries) && exact_applies(s, round_up(rows_kept, 64), k, entries);
  return by_exact || (rows_kept * list_div <= s->rows && scan_list_supported(s->nch, s->ld));
}
bool build_row_list(const Shard *s, const uint64_t *mask_words, int32_t n_tiles, int64_t rows_kept, int32_t k, int32_t entries,
                    std::vector<uint32_t> *ids) {
  if (!mask_words || !row_list_pays(s, rows_kept, k, entries)) return false;
  // (bits past the shard's last row cannot be set: slice_mask clears them; rows_kept is their exact count)
  ids->resize((size_t)round_up(rows_kept, 64) + 8);
  const size_t got = list_mask_bits(mask_words, n_tiles, rows_kept, ids->data());
  if (got == 0) return false;
  const size_t padded = (size_t)round_up((int64_t)got, 64);
  for (size_t i = got; i < padded; ++i) (*ids)[i] = 0xFFFFFFFFu;
  ids->resize(padded);
  return true;
}

int ctx_reserve_exact(Ctx *c, int64_t n) {
  if (n <= c->x_cap) return TSH_OK;
  hipFree(c->d_xkey);
  hipFree(c->d_xsum);
  c->d_xkey = nullptr;
  c->d_xsum = nullptr;
  c->bytes -= c->x_cap * 24;
  c->x_cap = 0;
  const int64_t want = std::min<int64_t>(round_up(n + n / 2, 1024), EX_MAX_ROWS);
  HIPCHK(hipMalloc(&c->d_xkey, (size_t)want * sizeof(uint64_t)));
  HIPCHK(hipMalloc(&c->d_xsum, (size_t)want * 2 * sizeof(double)));
  if (!c->d_xpick) {
    HIPCHK(hipMalloc(&c->d_xpick, (size_t)(1 + EX_MAX_ROWS / EX_R) * sizeof(uint64_t)));
    HIPCHK(hipMemset(c->d_xpick, 0, sizeof(uint64_t)));
    c->bytes += (1 + EX_MAX_ROWS / EX_R) * 8;
  }
  c->x_cap = want;
  c->bytes += want * 24;
  return TSH_OK;
}

// A search with at most TSH_OPT_EXACT_SCAN_ROWS rows to look at takes the exact path (tsh_exact.hip.h); its block must
// have room for the k rows it will hold
bool exact_applies(const Shard *s, int64_t n_exam, int32_t k, int32_t entries) {
  return n_exam > 0 && n_exam <= std::min<int64_t>(s->exact_rows, EX_MAX_ROWS) && k <= entries;
}
// E1's arguments but for where the query is (a.query / a.query_out: the caller's); q: the query, zero-padded to ld
void fill_exact_args(const Shard *s, const Ctx *c, bool use_list, bool dense_mask, int64_t n_exam, const float *q, ExactArgsQ *xa) {
  ExactArgs *a = &xa->a;
  a->rows = s->d_rows;
  a->query = c->d_query;
  a->query_out = nullptr;
  a->live = s->d_live;
  a->mask = dense_mask ? c->d_mask : nullptr;
  a->list = use_list ? c->d_list : nullptr;
  a->list_out = nullptr;
  a->xkey = c->d_xkey;
  a->xsum = c->d_xsum;
  a->wmin = nullptr;
  a->sqrt_mag_a = s->metric == TSH_METRIC_COSINE ? std::sqrt(query_mag_a(q, s->dim)) : 0.0;
  a->ld = s->ld;
  a->n_rows = s->rows;
  a->n_entries = (int32_t)n_exam;
  a->dim = s->dim;
}

// E1 of a short search (tsh_exact.hip.h): one wave per eight entries, the events on the kernel's own packet
void launch_exact_scan(const ExactArgsQ &xa, int metric, hipStream_t st, const LaunchEv &ev) {
  const unsigned grid = (unsigned)((xa.a.n_entries + EX_R - 1) / EX_R);
#define TSH_EXACT_LAUNCH(M)                                                                                      \
  do {                                                                                                           \
    if (ev.start || ev.stop) hipExtLaunchKernelGGL(exact_scan_kernel<M>, dim3(grid), dim3(64), 0, st, ev.start, ev.stop, 0, xa); \
    else exact_scan_kernel<M><<<grid, 64, 0, st>>>(xa);                                                          \
  } while (0)
  if (metric == TSH_METRIC_L2) TSH_EXACT_LAUNCH(METRIC_L2);
  else if (metric == TSH_METRIC_IP) TSH_EXACT_LAUNCH(METRIC_IP);
  else TSH_EXACT_LAUNCH(METRIC_COS);
#undef TSH_EXACT_LAUNCH
}

int job_enqueue(Shard *s, Job *j, const float *query, int32_t k, int32_t entries,
                const uint64_t *mask_words, uint64_t epoch, uint8_t *dev_target, int64_t rows_est = 0,
                const RowList *list = nullptr, bool more_coming = false, bool last_of_call = false, uint32_t tag = 0,
                const MaskPart *mp = nullptr) {
  // mp: the mask is a handle's part -- mask_words are its host words, its device words and (list->d_ids) its list
  // are read in place: nothing of the mask is copied or uploaded here
  Ctx *c = j->c;
  int rc = ctx_prepare(s, c, entries, mask_words != nullptr && !mp);
  if (rc) return rc;
  const bool use_list = list && (list->ids || list->d_ids) && mask_words;
  const bool own_list = use_list && !list->d_ids;  // the context's copy of a host-made list
  if (own_list && (rc = ctx_reserve_list(c, list->padded))) return rc;
  j->d_mask = mp ? mp->d_words : c->d_mask;
  j->d_list = use_list ? (own_list ? c->d_list : list->d_ids) : nullptr;
  j->list_tiles = use_list ? list->padded / 64 : 0;
  const int32_t n_tiles = (int32_t)((s->rows + 63) / 64);
  // A search with only a few thousand rows to look at (a selective mask's list, a small index or shard) takes their
  // exact sums directly and selects among the exact distances: two dispatches instead of three, no f32 keys, no band
  // (tsh_exact.hip.h).  The block must have room for the k rows it will hold.
  const int64_t n_exam = use_list ? (int64_t)list->padded : s->rows;
  const bool exact = exact_applies(s, n_exam, k, entries);
  j->exact = exact;
  if (exact && (rc = ctx_reserve_exact(c, n_exam))) return rc;
  j->k = k;
  j->entries = entries;
  j->user_mask = mask_words != nullptr;
  j->masked = j->user_mask || !s->all_live;
  j->dev_target = dev_target;
  j->quar_sel.clear();
  if (!s->quar_ids.empty()) {
    quarantine_select(s, mask_words, &j->quar_sel);
    if (!j->quar_sel.empty() && !dev_target && !c->h_quar) {
      HIPCHK(hipHostMalloc(&c->h_quar, QUARANTINE_MAX * sizeof(BlockEntry), hipHostMallocMapped));
      HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void **>(&c->h_quar_dev), c->h_quar, 0));
    }
  }
  uint8_t *dev_block = dev_target ? dev_target : c->d_block;  // where the device header lives
  // (a list scan needs the mask words on the device only where quarantined rows are matched against them)
  const bool upload_mask = mask_words && !mp && c->mask_epoch != epoch && (!use_list || (!j->quar_sel.empty() && dev_target));
  if (upload_mask) {
    memcpy(c->h_mask, mask_words, (size_t)n_tiles * 8);
    c->mask_epoch = epoch;
  }
  const bool upload_list = own_list && c->list_epoch != epoch;
  if (upload_list) {
    memcpy(c->h_list, list->ids, (size_t)list->padded * sizeof(uint32_t));
    c->list_epoch = epoch;
  }
  static thread_local ScanArgsQ sa;  // 4 KiB: keep it off the stack of deep callers
  static thread_local ExactArgsQ xa;
  const bool inline_q = s->ld <= SCAN_Q_INLINE;
  float *qdst = inline_q ? (exact ? xa.q : sa.q) : c->h_query;
  memcpy(qdst, query, (size_t)s->dim * sizeof(float));
  for (int64_t i = s->dim; i < s->ld; ++i) qdst[i] = 0.f;
  Band band;
  if (exact) {
    fill_exact_args(s, c, use_list, j->user_mask && !use_list, n_exam, qdst, &xa);
    xa.a.mask = (j->user_mask && !use_list) ? j->d_mask : nullptr;  // (the context's copy, or a handle's resident words)
    xa.a.list = j->d_list;
    xa.a.query = inline_q ? nullptr : c->d_query;
    // (only the quarantine kernels read the device copy of an inline query)
    xa.a.query_out = inline_q && !j->quar_sel.empty() ? c->d_query : nullptr;
    if (upload_list) {  // a new list: E1 reads it where it is (pinned host memory) and leaves the device copy -- a
      xa.a.list = c->h_list_dev;  // DMA packet in front of the scan cost a lone masked query ~8 us before anything ran
      xa.a.list_out = c->d_list;
    }
    // E2' (the wide pick) bounds the k-th key by the k-th smallest wave minimum: it needs clearly more waves than k
    j->picked = s->exact_pick && (n_exam + EX_R - 1) / EX_R >= 2 * (int64_t)k;
    if (j->picked) xa.a.wmin = c->d_xpick + 1;
    j->eps_rel = j->delta_abs = 0.f;
    j->force_all = false;
  } else {
    fill_scan_args(s, c, j->masked, j->user_mask, &sa);
    sa.a.mask = (j->masked && j->user_mask) ? j->d_mask : nullptr;
    if (use_list) {
      sa.a.list = j->d_list;
      sa.a.n_tiles = j->list_tiles;
    }
    band = compute_band(s, qdst);
    j->eps_rel = band.eps_rel;
    j->delta_abs = band.delta_abs;
    j->force_all = band.force_all != 0;
    if (inline_q) {
      sa.a.query = nullptr;        // read q[] from the kernel-argument segment ...
      sa.a.query_out = c->d_query;  // ... and leave a device copy for the rerank kernel
    }
  }
  SelectArgs se{};
  se.gmin = c->d_gmin;
  se.keys = c->d_keys;
  se.hdr = reinterpret_cast<BlockHeader *>(dev_block);
  se.hdr_host = reinterpret_cast<BlockHeader *>(c->h_block_dev);
  se.cand_rows = c->d_cand;
  se.n_tiles = use_list ? j->list_tiles : n_tiles;
  se.list = j->d_list;
  se.tag = tag;
  se.k = k;
  se.cand_cap = entries;
  se.eps_rel = band.eps_rel;
  se.delta_abs = band.delta_abs;
  se.force_all = band.force_all;
  se.metric = s->metric;
  se.row_base = s->row_base;
  se.shard_rows = s->rows;
  RerankArgs ra{};
  ra.rows = s->d_rows;
  ra.query = c->d_query;
  ra.cand_rows = c->d_cand;
  ra.count_ptr = &se.hdr->count;
  ra.out = reinterpret_cast<BlockEntry *>((dev_target ? dev_target : c->h_block_dev) + sizeof(BlockHeader));
  ra.ld = s->ld;
  ra.row_base = s->row_base;
  ra.dim = s->dim;
  ra.cap = entries;
  ra.metric = s->metric;
  // with other queries already in flight -- or the caller about to submit more (the first query of a multi-query
  // call: left alone it kept its select + re-rank on the pipeline stream, in front of the call's third scan) -- the
  // tail moves to the reserved CUs; a lone query keeps everything in order on one stream (no hand-off latency)
  const bool overlap = (s->inflight.fetch_add(1) > 0 || more_coming) && s->cu_split;
  j->counted = true;
  {
    std::lock_guard<std::mutex> lk(*s->scan_mu);
    hipStream_t ps = s->scan_stream;
    int which = -1;  // >= 0: one of the two scan streams of a small shard, booked in DeviceStreams::scan_out
    {
      // Between two scans on one in-order stream the GPU idles for about 13 us (drain, write-back,
      // ramp-up).  That is 3 % of a 1 M-row scan but 20 % of a 125 k-row one (a shard of an 8-GPU
      // index), so small shards alternate between two streams and the next scan's workgroups fill
      // in as the previous one drains (+12 % queries/s at 125 k and 250 k rows).  The two scans
      // then run side by side, so each one's own duration roughly doubles; large shards keep one
      // stream, where a scan's duration is its HBM time.  TSH_SCAN_STREAMS=1 / 2 forces either.
      static const int forced = probe_env("TSH_SCAN_STREAMS") ? atoi(probe_env("TSH_SCAN_STREAMS")) : 0;
      // (a selective row mask makes a big shard's scan just as short: count the rows it keeps)
      const int64_t tiles_read = rows_est > 0 ? std::min<int64_t>(n_tiles, (rows_est + 63) / 64) : n_tiles;
      static const int min_tiles = probe_env("TSH_TWO_STREAM_MIN_TILES") ? atoi(probe_env("TSH_TWO_STREAM_MIN_TILES")) : 0;
      const bool two = forced == 2 || (forced != 1 && tiles_read < SMALL_SHARD_TILES && tiles_read >= min_tiles);
      if (overlap && two && s->scan_stream2) {
        DeviceStreams *ds = s->dstreams;
        for (size_t i = 0; i < ds->scan_out.size();) {  // scans that have finished since the last look
          if (hipEventQuery(ds->scan_out[i].ev) != hipErrorNotReady) {
            ds->scans_out[ds->scan_out[i].which]--;
            ds->scan_out[i] = ds->scan_out.back();
            ds->scan_out.pop_back();
          } else {
            ++i;
          }
        }
        (void)hipGetLastError();  // (hipErrorNotReady is an answer, not an error)
        which = ds->scans_out[0] != ds->scans_out[1] ? (ds->scans_out[1] < ds->scans_out[0] ? 1 : 0) : (int)(s->scan_seq++ & 1);
        if (which) ps = s->scan_stream2;
      }
    }
    if (upload_mask)
      HIPCHK(hipMemcpyAsync(c->d_mask, c->h_mask, (size_t)n_tiles * 8, hipMemcpyHostToDevice, ps));
    if (upload_list && !exact)
      HIPCHK(hipMemcpyAsync(c->d_list, c->h_list, (size_t)list->padded * sizeof(uint32_t), hipMemcpyHostToDevice, ps));
    if (!inline_q)
      HIPCHK(hipMemcpyAsync(c->d_query, c->h_query, (size_t)s->ld * sizeof(float), hipMemcpyHostToDevice, ps));
    // (short scans as ONE dispatch -- every workgroup scans its tiles, the one drawing the last ticket selects and
    // re-ranks -- were built in round 2 and measured slower than these three launches: C1 45 vs 37 us per call,
    // keep-1 % masks 11.8 k vs 23.8 k queries/s.  Removed in round 3)
    j->timed = (s->c_scans.load() & 3) == 0;  // sample every 4th scan with timing events
    LaunchEv ev;  // start / stop ride on the scan's own packet: no barrier packets between scans
    if (j->timed) {
      ev.start = c->ev0;
      ev.stop = c->ev1;
    } else if (overlap) {
      ev.stop = c->ev_scanned;
    }
    if (exact) launch_exact_scan(xa, s->metric, ps, ev);
    else if (use_list) launch_scan_list(sa, s->nch, s->metric, ps, ev);
    else
      launch_scan(sa, s->nch, s->metric, j->masked, ps, ev,
                  j->masked && scan_mostly_live(rows_est > 0 ? rows_est : s->rows - s->deleted, s->rows));
    if (which >= 0 && ev.stop) {
      s->dstreams->scan_out.push_back({c, ev.stop, which});
      s->dstreams->scans_out[which]++;
    }
    hipStream_t ts = ps;
    // (the last query of a call: nothing follows its scan on that stream, so its tail stays there -- in order, without
    // the ~10 us of a cross-stream event hand-off in front of the select, on the step every caller waits for)
    // (the exact path's select stays on its scan stream, in order behind its short scan, while the scan is short: no
    // cross-stream hand-off, and the two scan streams alternate whole queries -- 64-query calls on 10 k x 128: 20.6 -> 16.2 us
    // per query, 2 k x 768: 22 -> 16.7, 8 k x 768: 21 -> 18.4; from ~8 k rows of 768 on the tail queues win again: 16 k x
    // 768 23 against 25; tools/r5_x_inorder.sh)
#ifndef TSH_X_INORDER_MAX
#define TSH_X_INORDER_MAX 6500000
#endif
    // (round 6: behind the wide pick -- 5-6 us, one workgroup per 256 entries -- the tail queues' 16 CUs and their
    // cross-stream hand-off, ~10 us in front of the pick under load, never pay: 10 k x 768 in 64-query calls 18.5-23.9 ->
    // 15.8-19.2 us per query, same box, alternating, tools/r6_inorder_ab.sh)
    const bool x_inorder = exact && (j->picked || n_exam * s->ld <= (int64_t)TSH_X_INORDER_MAX);
    if (overlap && !last_of_call && !x_inorder) {
      // (one tail queue serialises select + re-rank of consecutive queries: ~40 us per query, which is what short
      // scans -- selective masks, small shards -- were then limited by)
      static const bool one_tail = probe_env("TSH_ONE_TAIL") != nullptr && probe_env("TSH_ONE_TAIL")[0] == '1';
      // (a small shard's two scan streams each have their tail queue: the scans of one stream end a scan's length
      // apart, longer than a tail, so no tail queues behind another -- alternating blindly put two tails of
      // near-simultaneous scan ends on one queue at the end of a call, 40 us in full view)
      if (which >= 0 && !one_tail && s->tail_stream2) ts = which ? s->tail_stream2 : s->tail_stream;
      else ts = (!one_tail && s->tail_stream2 && (s->tail_seq++ & 1)) ? s->tail_stream2 : s->tail_stream;
      HIPCHK(hipStreamWaitEvent(ts, ev.stop, 0));
    }
    // (short rows and lists, config C1: K2 + K4 as ONE dispatch, the selecting workgroup re-ranking its dozen
    // candidates a lane each, was tried -- 17 us against 9 + 4.4 for the two launches: the lone workgroup waits out
    // count -> candidate ids -> rows one after the other, which the second launch's ramp-up hides)
    j->last_stream = ts;
    j->enq_seq = ++s->dstreams->enq_counter;
    // the completion event rides on the last kernel's own dispatch packet unless more kernels follow (a separate
    // hipEventRecord is one more runtime call and one more barrier packet per query)
    bool done_recorded = false;
    if (exact) {
      ExactSelArgs xs{};
      xs.xkey = c->d_xkey;
      xs.xsum = c->d_xsum;
      xs.list = j->d_list;
      xs.hdr = se.hdr;
      xs.hdr_host = se.hdr_host;
      xs.out = ra.out;
      xs.row_base = s->row_base;
      xs.shard_rows = s->rows;
      xs.n_entries = xa.a.n_entries;
      xs.k = k;
      xs.cap = entries;
      xs.metric = s->metric;
      xs.tag = tag;
      j->xsel = xs;
      if (j->picked) {  // E2': one workgroup per 256 entries, a bound from E1's wave minima, no ranking below it
        ExactPickArgs xp{};
        xp.xkey = xs.xkey;
        xp.xsum = xs.xsum;
        xp.list = xs.list;
        xp.wmin = c->d_xpick + 1;
        xp.ctr = reinterpret_cast<unsigned long long *>(c->d_xpick);
        xp.hdr = xs.hdr;
        xp.hdr_host = xs.hdr_host;
        xp.out = xs.out;
        xp.row_base = xs.row_base;
        xp.shard_rows = xs.shard_rows;
        xp.n_entries = xs.n_entries;
        xp.n_groups = (xs.n_entries + EX_R - 1) / EX_R;
        xp.k = k;
        xp.cap = entries;
        xp.metric = s->metric;
        xp.tag = tag;
        const unsigned pg = (unsigned)((xs.n_entries + 255) / 256);
        if (j->quar_sel.empty()) {
          hipExtLaunchKernelGGL(exact_pick_kernel, dim3(pg), dim3(256), 0, ts, nullptr, c->ev_done, 0, xp);
          done_recorded = true;
        } else {
          exact_pick_kernel<<<pg, 256, 0, ts>>>(xp);
        }
      } else if (j->quar_sel.empty()) {
        hipExtLaunchKernelGGL(exact_select_kernel, dim3(1), dim3(1024), 0, ts, nullptr, c->ev_done, 0, xs);
        done_recorded = true;
      } else {
        exact_select_kernel<<<1, 1024, 0, ts>>>(xs);
      }
    } else {
      launch_select(se, se.n_tiles, ts);
      if (j->quar_sel.empty()) {
        hipExtLaunchKernelGGL(rerank_kernel, dim3((unsigned)std::min(entries, 1024)), dim3(64), 0, ts, nullptr, c->ev_done, 0, ra);
        done_recorded = true;
      } else {
        rerank_kernel<<<std::min(entries, 1024), 64, 0, ts>>>(ra);
      }
    }
    if (!j->quar_sel.empty() && dev_target) {
      launch_quarantine_append(s, c, j, ts);
    } else if (!j->quar_sel.empty()) {
      QuarArgs qa{};
      qa.rows = s->d_rows;
      qa.Q = c->d_query;
      qa.list = s->d_quar;
      qa.out = c->h_quar_dev;
      qa.ld = s->ld;
      qa.ldq = s->ld;
      qa.row_base = s->row_base;
      qa.dim = s->dim;
      qa.cap = (int32_t)QUARANTINE_MAX;
      qa.metric = s->metric;
      quarantine_kernel<<<dim3((unsigned)((s->quar_ids.size() + 63) / 64), 1), 64, 0, ts>>>(qa);
    }
    if (!done_recorded) HIPCHK(hipEventRecord(c->ev_done, ts));
  }
  s->c_scans++;
  if (use_list) s->c_list_scans++;
  if (exact) s->c_exact_scans++;
  return TSH_OK;
}

// Wide-band path for a query whose K2 list overflowed (ties / degenerate data):
// whole-grid filter of this context's keys[] + f64 rerank of everything in the band.
// host copies of f2key / key2f / band_of (tsh_kernels.hip.h)
inline uint32_t h_f2key(float f) {
  if (f != f) return KEY_NAN;
  uint32_t b;
  memcpy(&b, &f, 4);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
inline float h_key2f(uint32_t k) {
  uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  float f;
  memcpy(&f, &b, 4);
  return f;
}
inline uint32_t h_band_of(uint32_t tau_key, float eps_rel, float delta_abs) {
  if (tau_key >= KEY_NAN) return KEY_NAN;
  float t = h_key2f(tau_key);
  if (t == INFINITY) return KEY_NAN;
  double w = (double)t + std::fabs((double)t) * (double)eps_rel + (double)delta_abs;
  float f = (float)w;
  if ((double)f < w) f = std::nextafter(f, INFINITY);
  if (!(f < INFINITY)) return KEY_NAN;
  return h_f2key(f);
}

int run_fallback(Shard *s, Job *j, uint32_t band_key, std::vector<BlockEntry> *spill) {
  Ctx *c = j->c;
  hipStream_t st = s->aux_stream;
  // (a list scan left its keys in list order: fewer keys, and the filter maps positions back to row ids)
  const int64_t n_keys = j->list_tiles > 0 ? (int64_t)j->list_tiles * 64 : ((s->rows + 63) / 64) * 64;
  const uint32_t *list = j->list_tiles > 0 ? j->d_list : nullptr;
  if (s->cap > c->big_cap) {
    hipFree(c->d_big_rows);
    hipFree(c->d_big_entries);
    c->d_big_rows = nullptr;
    c->d_big_entries = nullptr;
    HIPCHK(hipMalloc(&c->d_big_rows, (size_t)s->cap * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&c->d_big_entries, (size_t)s->cap * sizeof(BlockEntry)));
    c->bytes += (s->cap - c->big_cap) * 28;
    c->big_cap = s->cap;
  }
  int fgrid = (int)std::min<int64_t>((n_keys + 255) / 256, 4096);
  if (band_key >= KEY_NAN && !j->force_all && j->k < n_keys) {
    // K2 could not bound the k-th key (k beyond its tile-minimum scheme, or fewer than k live tiles and a full
    // list): find the exact k-th smallest key with a 4-pass radix select over all keys -- 4 small kernels and
    // host round trips instead of an f64 rerank of every row (k = 2000 on 1 M rows: 0.9 ms instead of 9)
    if (!c->d_hist) HIPCHK(hipMalloc(&c->d_hist, 256 * sizeof(uint32_t)));
    uint32_t prefix = 0, hist[256];
    uint64_t remaining = (uint64_t)j->k;
    bool found = true;
    for (int shift = 24; shift >= 0 && found; shift -= 8) {
      HIPCHK(hipMemsetAsync(c->d_hist, 0, sizeof hist, st));
      radix_hist_kernel<<<fgrid, 256, 0, st>>>(c->d_keys, c->d_gmin, n_keys, prefix, shift, c->d_hist);
      HIPCHK(hipMemcpyAsync(hist, c->d_hist, sizeof hist, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      uint64_t cum = 0;
      int b = 0;
      for (; b < 256; ++b) {
        if (cum + hist[b] >= remaining) break;
        cum += hist[b];
      }
      if (b == 256) {
        found = false;  // fewer than k live keys: everything is a candidate
      } else {
        remaining -= cum;
        prefix |= (uint32_t)b << shift;
      }
    }
    if (found) band_key = h_band_of(prefix, j->eps_rel, j->delta_abs);
  }
  HIPCHK(hipMemsetAsync(c->d_big_count, 0, 4, st));
  filter_kernel<<<fgrid, 256, 0, st>>>(c->d_keys, c->d_gmin, n_keys, band_key, c->d_big_rows, c->d_big_count,
                                      (uint32_t)c->big_cap, list);
  uint32_t count = 0;
  HIPCHK(hipMemcpyAsync(&count, c->d_big_count, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  RerankArgs ra{};
  ra.rows = s->d_rows;
  ra.query = c->d_query;
  ra.cand_rows = c->d_big_rows;
  ra.count_ptr = c->d_big_count;
  ra.out = c->d_big_entries;
  ra.ld = s->ld;
  ra.row_base = s->row_base;
  ra.dim = s->dim;
  ra.cap = (int32_t)std::min<int64_t>(c->big_cap, 0x7FFFFFFF);
  ra.metric = s->metric;
  if (count > 0) {
    int rgrid = (int)std::min<uint32_t>(count, 16384u);
    rerank_kernel<<<rgrid, 64, 0, st>>>(ra);
  }
  BlockHeader *h = reinterpret_cast<BlockHeader *>(c->h_block);
  h->count = count;
  h->flags = count > (uint32_t)j->entries ? FLAG_LIST_OVERFLOW : 0u;
  HIPCHK(hipMemcpyAsync(j->dev_target ? j->dev_target : c->d_block, h, sizeof *h, hipMemcpyHostToDevice, st));
  uint32_t fit = std::min<uint32_t>(count, (uint32_t)j->entries);
  if (fit) {
    if (j->dev_target)
      HIPCHK(hipMemcpyAsync(j->dev_target + sizeof(BlockHeader), c->d_big_entries, (size_t)fit * sizeof(BlockEntry),
                            hipMemcpyDeviceToDevice, st));
    else
      HIPCHK(hipMemcpyAsync(c->h_block + sizeof(BlockHeader), c->d_big_entries, (size_t)fit * sizeof(BlockEntry),
                            hipMemcpyDeviceToHost, st));
  }
  if (spill && count > (uint32_t)j->entries) {
    spill->resize(count);
    HIPCHK(hipMemcpyAsync(spill->data(), c->d_big_entries, (size_t)count * sizeof(BlockEntry),
                          hipMemcpyDeviceToHost, st));
  }
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  s->c_fallbacks++;
  s->c_cands += count;
  return TSH_OK;
}

// Waits for a job; afterwards c->h_block / c->d_block hold the final block
// (and *spill every candidate when they did not fit); *extra gets the quarantined rows' entries.
int job_finish(Shard *s, Job *j, std::vector<BlockEntry> *spill, std::vector<BlockEntry> *extra) {
  Ctx *c = j->c;
  if (j->counted) {
    s->inflight.fetch_sub(1);
    j->counted = false;
  }
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipEventSynchronize(c->ev_done));
  HIPCHK(hipGetLastError());
  if (j->timed) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) {
      std::lock_guard<std::mutex> lk(s->ctx_mu);
      s->scan_us_sum += (double)ms * 1e3;
      s->scan_us_samples++;
    }
    j->timed = false;
  }
  BlockHeader *h = reinterpret_cast<BlockHeader *>(c->h_block);
#ifdef TSH_PROBES
  if (j->exact && probe_env("TSH_X2_TRACE"))
    fprintf(stderr, "[x2] keys %.2f select %.2f entries %.2f us, %u histogram rounds, %u ranked, %u out adds %.2f scan %.2f\n", h->tau_key * 0.01,
            h->band_key * 0.01, h->tiles_hit * 0.01, h->pad[2], h->pad[3], h->count, (h->pad[0] >> 16) * 0.01, (h->pad[0] & 0xFFFF) * 0.01);
#endif
  if ((h->flags & FLAG_LIST_OVERFLOW) && j->leave_overflow && j->dev_target) {
    s->c_cands += std::min(h->count, h->entries);  // (the block stays as it is: see Job::leave_overflow)
  } else if ((h->flags & FLAG_LIST_OVERFLOW) && j->exact) {
    // The wide pick emits every row up to its cut bin: ties by the hundred, or a k-th neighbour outside the histogram's
    // window, and the bin holds more rows than the block.  The keys and sums of all entries are still in the context:
    // exact_select_kernel ranks them and writes exactly min(k, live rows) entries -- into a block exact_applies sized for
    // k, so THAT cannot overflow (should the invariant ever slip: an error, never the wide-band pass, for which the
    // context holds no f32 keys).
    if (!j->picked) return set_err(TSH_E_HIP, "the exact path's block overflowed (%u of %u entries): internal error", h->count, h->entries);
    hipStream_t st = s->aux_stream;
    exact_select_kernel<<<1, 1024, 0, st>>>(j->xsel);
    if (!j->quar_sel.empty() && j->dev_target) launch_quarantine_append(s, c, j, st);  // (the block's count was rewritten)
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    s->c_pick_redone++;
    if (h->flags & FLAG_LIST_OVERFLOW)
      return set_err(TSH_E_HIP, "the exact path's block overflowed (%u of %u entries): internal error", h->count, h->entries);
    s->c_cands += h->count;
  } else if (h->flags & FLAG_LIST_OVERFLOW) {
    int rc = run_fallback(s, j, h->band_key, spill);  // keys[] of this query are still in the context
    if (rc) return rc;
    if (!j->quar_sel.empty() && j->dev_target) {  // the fallback rewrote the device block: append again
      launch_quarantine_append(s, c, j, s->aux_stream);
      HIPCHK(hipStreamSynchronize(s->aux_stream));
    }
  } else {
    s->c_cands += h->count;
  }
  if (!j->quar_sel.empty() && !j->dev_target) {  // the quarantined rows join the candidates
    if (!extra) return set_err(TSH_E_BAD_ARG, "no room for the quarantined rows' entries");
    extra->clear();
    for (uint32_t i : j->quar_sel) extra->push_back(c->h_quar[i]);
  }
  s->c_searches++;
  return TSH_OK;
}

struct SearchOut {
  uint8_t *h_blocks = nullptr;  // host mode: nq blocks (caller memory)
  std::vector<std::vector<BlockEntry>> *spill = nullptr;
  std::vector<std::vector<BlockEntry>> *extra = nullptr;  // per query: entries of the quarantined rows
  int32_t q_base = 0;  // batched path: this part's query 0 is (*extra)[q_base]
  uint8_t *d_blocks = nullptr;  // device mode
  hipStream_t user_stream = nullptr;
  // batched path, host mode: called as soon as the blocks of queries [q0, q1) have arrived from the GPU, while it
  // still re-ranks the next chunk; query q's block is at base + q * block bytes (the batch's own pinned buffer:
  // valid during the call only -- with a callback the blocks are NOT copied to h_blocks); skip[q] != 0 marks
  // queries the single-query path will redo (those do land in h_blocks)
  // (mag_a[q]: query_mag_a of query q, computed while the queries were prepared)
  std::function<void(int32_t q0, int32_t q1, const char *skip, const uint8_t *base, const double *mag_a)> on_chunk;
  // batched path, host mode: the caller takes FINAL results (threshold fin_thr applied, ordered, cut to k) -- the
  // device finalises whatever it can (rerank_final_kernel) and hands over ids / distances (k per query) and counts of
  // queries [q0, q1), indexed like the call's queries; what it cannot goes through on_chunk as before
  double fin_thr = std::nan("");
  std::function<void(int32_t q0, int32_t q1, const char *skip, const int64_t *ids, const double *dist, const int32_t *cnt)> on_final;
  // single-query pipeline, device mode (progressive shard search): called once query q's block is final in
  // d_blocks (its job finished on the host, fallback included), from whichever submitting thread ran it.  With it
  // set the submitting threads take the queries interleaved (q = t, t + T, ...), so blocks become final in order
  std::function<void(int32_t q)> on_done;
  // ... and once query q's kernels are all ENQUEUED: `done` is the event behind the last writer of its block
  // (`where`: the stream that event was recorded on -- kernels of one stream finish in order; `seq`: the job's place
  // in the order the device's streams were fed, which with two submitting threads is not the queries' order)
  std::function<void(int32_t q, hipEvent_t done, hipStream_t where, uint64_t seq)> on_enqueued;
  uint32_t tag = 0;  // generation stamped into the blocks' headers (BlockHeader.pad[1])
  bool leave_overflow = false;  // device mode: see Job::leave_overflow
};

// One submitting thread's share of a multi-query call: queries [q0, q1) of the call, at most
// `depth` of them in flight on separate contexts so one query's select / rerank / copies hide
// behind the next query's scan.
int shard_search_slice(Shard *s, const float *queries, int32_t q0, int32_t q1, int32_t k, const uint64_t *mask_words,
                       uint64_t epoch, int32_t entries, SearchOut *out, int depth, int64_t rows_est,
                       const RowList *list = nullptr, int32_t stride = 1, const MaskPart *mp = nullptr) {
  // this thread's queries: q0, q0 + stride, ... below q1 (cnt of them; i-th = q0 + i * stride)
  const size_t bb = (size_t)tsh_candidate_block_bytes(entries);
  const int32_t cnt = q1 > q0 ? (q1 - q0 + stride - 1) / stride : 0;
  depth = std::max(1, std::min(depth, std::min(cnt, MAX_CTX)));
  std::vector<Job> jobs((size_t)depth);
  int rc = TSH_OK;
  int32_t submitted = 0, finished = 0;
  auto release_all = [&]() {
    for (auto &j : jobs)
      if (j.c) {
        if (j.counted) hipEventSynchronize(j.c->ev_done);
        if (j.counted) s->inflight.fetch_sub(1);
        j.counted = false;
        ctx_release(s, j.c);
        j.c = nullptr;
      }
  };
  while (finished < cnt) {
    while (submitted < cnt && submitted - finished < depth) {
      // wait for a context only while holding none: callers that each hold some and wait for
      // more would deadlock on the shard's fixed pool
      Ctx *c = ctx_acquire(s, submitted == finished);
      if (!c) break;
      Job &j = jobs[(size_t)(submitted % depth)];
      j.c = c;
      j.leave_overflow = out->leave_overflow;
      const int32_t q = q0 + submitted * stride;
      rc = job_enqueue(s, &j, queries + (size_t)q * s->dim, k, entries, mask_words, epoch,
                       out->d_blocks ? out->d_blocks + (size_t)q * bb : nullptr, rows_est, list, cnt > 1,
                       cnt > 1 && submitted == cnt - 1, out->tag, mp);
      if (rc) {
        release_all();
        return rc;
      }
      ++submitted;
      if (out->on_enqueued) out->on_enqueued(q, c->ev_done, j.last_stream, j.enq_seq);
    }
    Job &j = jobs[(size_t)(finished % depth)];
    const int32_t q = q0 + finished * stride;
    std::vector<BlockEntry> *sp = out->spill ? &(*out->spill)[(size_t)q] : nullptr;
    std::vector<BlockEntry> *ex = out->extra ? &(*out->extra)[(size_t)q] : nullptr;
    rc = job_finish(s, &j, sp, ex);
    if (rc) {
      release_all();
      return rc;
    }
    if (out->h_blocks) memcpy(out->h_blocks + (size_t)q * bb, j.c->h_block, bb);
    ctx_release(s, j.c);
    j.c = nullptr;
    ++finished;
    if (out->on_done) out->on_done(q);
  }
  return TSH_OK;
}

// nq single-query searches.  Each query costs the host about seven runtime calls (three
// launches, events, a wait); on a small shard (a row range of a multi-GPU index) that is
// longer than the scan itself, so larger calls are submitted from a few threads at once.
// Caller holds s->mu shared.
int shard_search_blocks(Shard *s, const float *queries, int32_t nq, int32_t k, const MaskSrc &mask,
                        int32_t entries, SearchOut *out, int depth) {
  const int32_t n_tiles = (int32_t)((s->rows + 63) / 64);
  // (the calling thread's buffers, kept between calls: a lone masked query does not pay for two allocations)
  static thread_local std::vector<uint64_t> mask_words;
  static thread_local std::vector<uint32_t> list_ids;
  uint64_t epoch = 0;
  int64_t rows_est = 0;
  const MaskPart *mp = mask.part;
  RowList list;
  if (mp) {  // a handle: sliced, counted and listed when it was made -- nothing to do per call
    rows_est = std::max<int64_t>(mp->kept, 1);
    if (mp->list_padded > 0 && row_list_pays(s, mp->kept, k, entries)) {
      list.d_ids = mp->d_list;
      list.padded = mp->list_padded;
    }
  } else if (mask) {
    if (mask_words.size() < (size_t)n_tiles) mask_words.resize((size_t)n_tiles);
    slice_mask(s, mask.bytes, mask_words.data(), n_tiles);
    epoch = s->mask_epoch_src.fetch_add(1);
    // one pass over the mask per call: how long will each scan be?  (decides one- or two-stream pipelining)
    rows_est = popcount_words(mask_words.data(), (size_t)n_tiles);
    if (rows_est == 0) rows_est = 1;
    // (made here once for all queries of the call)
    if (build_row_list(s, mask_words.data(), n_tiles, rows_est, k, entries, &list_ids)) {
      list.ids = list_ids.data();
      list.padded = (int32_t)list_ids.size();
    }
  }
  const RowList *lp = (list.ids || list.d_ids) ? &list : nullptr;
  const uint64_t *mw = mp ? mp->h_words.data() : (mask ? mask_words.data() : nullptr);
  // short scans (small shards, selective masks) are bound by the submitting thread's ~25 us per query: two threads
  const int64_t scan_bytes = (rows_est > 0 ? rows_est : s->rows) * s->ld * 4;
  static const int forced_threads = probe_env("TSH_SUBMIT_THREADS") ? atoi(probe_env("TSH_SUBMIT_THREADS")) : 0;
  const int want = forced_threads > 0 ? forced_threads : (scan_bytes <= (160ll << 20) ? 2 : SUBMIT_THREADS);
  const int T = std::min(want, nq / 8);
  if (T <= 1) return shard_search_slice(s, queries, 0, nq, k, mw, epoch, entries, out, depth, rows_est, lp, 1, mp);
  std::vector<int> rcs((size_t)T, TSH_OK);
  std::vector<std::string> errs((size_t)T);
  const int per_depth = std::max(2, depth / T);
  const bool interleave = (bool)out->on_done;  // a progressive search wants its blocks final in query order
  auto run = [&](int t) {
    const int32_t q0 = interleave ? t : (int32_t)((int64_t)nq * t / T);
    const int32_t q1 = interleave ? nq : (int32_t)((int64_t)nq * (t + 1) / T);
    rcs[(size_t)t] = shard_search_slice(s, queries, q0, q1, k, mw, epoch, entries, out, per_depth, rows_est, lp,
                                        interleave ? T : 1, mp);
    if (rcs[(size_t)t]) errs[(size_t)t] = g_err;
  };
  std::vector<std::thread> th;
  for (int t = 1; t < T; ++t) th.emplace_back(run, t);
  run(0);
  for (auto &x : th) x.join();
  for (int t = 0; t < T; ++t)
    if (rcs[(size_t)t]) {
      g_err = errs[(size_t)t];
      return rcs[(size_t)t];
    }
  return TSH_OK;
}

#include "tsh_host_batch.inl.h"  // batched (matrix-core) path, host side

}  // namespace

// ============================================================================
// an un-waited asynchronous search (tsh_search_submit): one job per shard; the
// shards stay share-locked until the wait so appends cannot move the rows
struct Ticket {
  int32_t k = 0, entries = 0;
  std::vector<float> query;
  std::vector<Job> jobs;
  std::vector<std::shared_lock<RwLock>> locks;
};

struct tsh_index {
  int dim = 0, metric = 0;
  int64_t rows_per_shard = 0;  // multi-device split (0: single shard)
  std::vector<std::unique_ptr<Shard>> shards;
  std::mutex mu;  // serialises append routing
  std::mutex tk_mu;
  std::vector<std::unique_ptr<Ticket>> tickets;
  std::atomic<int> tickets_open{0};  // submitted, not yet waited for: each holds one context per shard
  std::atomic<int32_t> batch_min_nq{1};  // 0 never, 1 by estimated cost, n >= 2: from n queries per call on
  std::unique_ptr<ShardWorkers> workers;  // multi-device handles: one persistent host thread per further shard
  // result-block buffers of multi-query calls, kept between calls: a fresh 6 MB allocation per call spends
  // ~0.3 ms in page faults when it is first written
  std::mutex pool_mu;
  std::vector<std::pair<size_t, std::unique_ptr<uint8_t[]>>> block_pool;
  std::unique_ptr<uint8_t[]> take_blocks(size_t bytes, size_t *cap) {
    std::lock_guard<std::mutex> lk(pool_mu);
    for (size_t i = 0; i < block_pool.size(); ++i)
      if (block_pool[i].first >= bytes) {
        *cap = block_pool[i].first;
        std::unique_ptr<uint8_t[]> p = std::move(block_pool[i].second);
        block_pool.erase(block_pool.begin() + (long)i);
        return p;
      }
    *cap = bytes;
    return std::unique_ptr<uint8_t[]>(new uint8_t[bytes]);
  }
  void give_blocks(std::unique_ptr<uint8_t[]> p, size_t cap) {
    std::lock_guard<std::mutex> lk(pool_mu);
    if (block_pool.size() < 4 && cap <= ((size_t)64 << 20)) block_pool.emplace_back(cap, std::move(p));
  }
};

namespace {

// Shared lock of a shard on behalf of a call on `idx`.  A handle with asynchronous tickets open passes a
// waiting writer (RwLock: queueing behind it would deadlock ticket holder, writer and this call).
inline std::shared_lock<RwLock> share(tsh_index *idx, Shard *s) {
  s->mu.lock_shared_gate(idx->tickets_open.load() > 0);
  return std::shared_lock<RwLock>(s->mu, std::adopt_lock);
}

int make_shard(int dim, int metric, int device, int64_t row_base, int64_t cap_rows,
               std::unique_ptr<Shard> *out) {
  std::unique_ptr<Shard> s(new Shard());
  s->device = device;
  s->dim = dim;
  s->metric = metric;
  s->ld = round_up(dim, 4);
  s->nch = pick_nch((int)(s->ld / 4));
  s->row_base = row_base;
  int rc = shard_init(s.get());
  if (rc) return rc;
  s->batch = new BatchCtx();
  s->batch2 = new BatchCtx();
  if (cap_rows > 0) {
    rc = shard_reserve(s.get(), cap_rows);
    if (rc) return rc;
  }
  *out = std::move(s);
  return TSH_OK;
}

void shard_destroy(Shard *s) {
  hipSetDevice(s->device);
  for (auto &c : s->ctx_all) ctx_free_all(c.get());
  for (BatchCtx **b : {&s->batch, &s->batch2})
    if (*b) {
      batch_free(*b);
      delete *b;
      *b = nullptr;
    }
  // (streams are the device's, not the shard's; the last shard on a device takes the CU-masked ones along)
  hipFree(s->d_rows);
  hipFree(s->d_inv_norm);
  hipFree(s->d_sqnorm);
  hipFree(s->d_live);
  hipFree(s->d_split);
  hipFree(s->d_perm);
  hipFree(s->d_psq);
  hipFree(s->d_hub);
  hipFree(s->d_hub_ids);
  hipFree(s->d_hub_sq);
  hipFree(s->d_stats);
  hipFree(s->d_tmp_u32);
  hipFree(s->d_del_ids);
  hipFree(s->d_quar);
  hipFree(s->d_irr);
  if (s->holds_streams) {
    s->holds_streams = false;
    device_streams_release(s->device);
  }
}

int check_create_args(int32_t dim, int32_t metric, int64_t cap, tsh_index **out) {
  if (!out) return set_err(TSH_E_BAD_ARG, "out is NULL");
  *out = nullptr;
  if (dim <= 0 || dim > MAX_DIM_SCAN)
    return set_err(TSH_E_BAD_ARG, "dim %d outside [1,%d]", dim, MAX_DIM_SCAN);
  if (metric < 0 || metric > 2) return set_err(TSH_E_BAD_ARG, "metric %d unknown", metric);
  if (cap < 0) return set_err(TSH_E_BAD_ARG, "capacity_rows < 0");
  if (device_count_cached() <= 0) return set_err(TSH_E_NO_DEVICE, "no HIP device available");
  return TSH_OK;
}

Shard *shard_for_row(tsh_index *idx, int64_t gid) {
  if (idx->shards.size() == 1) return idx->shards[0].get();
  int64_t g = idx->rows_per_shard > 0 ? gid / idx->rows_per_shard : 0;
  if (g >= (int64_t)idx->shards.size()) g = (int64_t)idx->shards.size() - 1;
  if (g < 0) g = 0;
  return idx->shards[(size_t)g].get();
}

int index_append(tsh_index *idx, int64_t first, int64_t n, const float *rows, bool dev) {
  if (!idx) return set_err(TSH_E_BAD_ARG, "index is NULL");
  if (n < 0 || first < 0) return set_err(TSH_E_BAD_ARG, "negative row range");
  if (n == 0) return TSH_OK;
  if (!rows) return set_err(TSH_E_BAD_ARG, "rows is NULL");
  if (dev && idx->shards.size() > 1)
    return set_err(TSH_E_BAD_ARG, "append_device needs a single-device handle");
  std::lock_guard<std::mutex> lk(idx->mu);
  int64_t done = 0;
  while (done < n) {
    int64_t gid = first + done;
    Shard *s = shard_for_row(idx, gid);
    int64_t take = n - done;
    if (idx->shards.size() > 1 && s != idx->shards.back().get()) {
      int64_t end = s->row_base + idx->rows_per_shard;
      take = std::min(take, end - gid);
    }
    if (gid < s->row_base) return set_err(TSH_E_BAD_ARG, "row id %lld below shard base", (long long)gid);
    std::unique_lock<RwLock> xl(s->mu);
    int rc = shard_append(s, gid - s->row_base, take, rows + (size_t)done * idx->dim, dev);
    if (rc) return rc;
    done += take;
  }
  return TSH_OK;
}

}  // namespace

// ---- mask handles ---------------------------------------------------------------------------------------------
struct tsh_mask {
  tsh_index *idx = nullptr;
  std::vector<uint8_t> bits;  // the caller's GLOBAL bitmap (its own copy; zero-extended as the index grows)
  std::mutex mu;              // builds / rebuilds of the parts
  std::vector<std::unique_ptr<MaskPart>> parts;  // one per shard of the index
};

namespace {

void mask_part_free(MaskPart *p) {
  if (hipSetDevice(p->device) != hipSuccess) return;
  hipFree(p->d_words);
  hipFree(p->d_list);
  hipFree(p->d_bsum);
  p->d_words = nullptr;
  p->d_list = p->d_bsum = nullptr;
  p->words_cap = p->list_cap = p->bsum_cap = 0;
  p->bytes = 0;
}

// (Re)builds one shard's part for the rows the shard has now.  Caller holds s->mu shared (the rows cannot change) and
// m->mu.  The words are sliced and counted on the host once -- the batched path and the quarantined rows read them
// there --, uploaded once, and a selective mask's list is compacted on the device (tsh_mask.hip.h).
int mask_build_part(tsh_mask *m, MaskPart *p, Shard *s) {
  HIPCHK(hipSetDevice(s->device));
  p->device = s->device;
  const int64_t rows = s->rows;
  const int32_t n_tiles = (int32_t)((rows + 63) / 64);
  // bits at or beyond the caller's n_bytes are not kept: the handle's copy grows with zeros to what slice_mask reads
  const size_t need = (size_t)((s->row_base + rows + 7) / 8) + 1;
  if (m->bits.size() < need) m->bits.resize(need, 0);
  p->h_words.assign((size_t)n_tiles, 0);
  if (n_tiles > 0) slice_mask(s, m->bits.data(), p->h_words.data(), n_tiles);
  p->pre.assign((size_t)n_tiles + 1, 0);
  for (int32_t t = 0; t < n_tiles; ++t) p->pre[(size_t)t + 1] = p->pre[(size_t)t] + __builtin_popcountll(p->h_words[(size_t)t]);
  p->kept = p->pre[(size_t)n_tiles];
  p->n_tiles = n_tiles;
  p->list_padded = 0;
  hipStream_t st = s->aux_stream;
  if (n_tiles > p->words_cap) {
    hipFree(p->d_words);
    p->d_words = nullptr;
    p->bytes -= p->words_cap * 8;
    p->words_cap = 0;
    const int64_t want = round_up(n_tiles + n_tiles / 8, 64);
    HIPCHK(hipMalloc(&p->d_words, (size_t)want * 8));
    p->words_cap = want;
    p->bytes += want * 8;
  }
  if (n_tiles > 0)
    HIPCHK(hipMemcpyAsync(p->d_words, p->h_words.data(), (size_t)n_tiles * 8, hipMemcpyHostToDevice, st));
  // the list: wherever a search may scan the kept rows as one (row_list_pays: few enough for the exact path at any
  // k and row width, or fewer than one row in list_div) -- compacted on the device, in id order
  const int64_t padded = round_up(p->kept, 64);
  const int64_t list_div = mask_list_div();
  const bool want_list = p->kept > 0 && rows >= 4096 && list_div > 0 && padded < 0x7FFFFFC0ll &&
                         (padded <= EX_MAX_ROWS || p->kept * list_div <= rows);
  if (want_list) {
    const int32_t n_blocks = (n_tiles + MASK_BLOCK_WORDS - 1) / MASK_BLOCK_WORDS;
    if (n_blocks + 1 > p->bsum_cap) {
      hipFree(p->d_bsum);
      p->d_bsum = nullptr;
      p->bytes -= p->bsum_cap * 4;
      p->bsum_cap = 0;
      const int64_t want = round_up(n_blocks + 1 + n_blocks / 8, 64);
      HIPCHK(hipMalloc(&p->d_bsum, (size_t)want * 4));
      p->bsum_cap = want;
      p->bytes += want * 4;
    }
    if (padded > p->list_cap) {
      hipFree(p->d_list);
      p->d_list = nullptr;
      p->bytes -= p->list_cap * 4;
      p->list_cap = 0;
      const int64_t want = round_up(padded + padded / 8, 1024);
      HIPCHK(hipMalloc(&p->d_list, (size_t)want * 4));
      p->list_cap = want;
      p->bytes += want * 4;
    }
    uint32_t *d_total = p->d_bsum + n_blocks;
    mask_block_count_kernel<<<n_blocks, MASK_BLOCK_WORDS, 0, st>>>(p->d_words, n_tiles, p->d_bsum);
    mask_compact_kernel<<<n_blocks, MASK_BLOCK_WORDS, 0, st>>>(p->d_words, n_tiles, p->d_bsum, p->d_list, d_total);
    if (padded > p->kept)
      HIPCHK(hipMemsetAsync(p->d_list + p->kept, 0xFF, (size_t)(padded - p->kept) * 4, st));
    uint32_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, d_total, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    if ((int64_t)total != p->kept)  // (the host counted the same words: cannot differ)
      return set_err(TSH_E_HIP, "mask compaction listed %u rows, the mask keeps %lld", total, (long long)p->kept);
    p->list_padded = (int32_t)padded;
  } else {
    HIPCHK(hipStreamSynchronize(st));
  }
  p->built_rows.store(rows, std::memory_order_release);
  return TSH_OK;
}

// shard g's part of a handle, current for the rows the shard has (caller holds s->mu shared); nullptr + *rc on failure
const MaskPart *mask_part(tsh_mask *m, size_t g, Shard *s, int *rc) {
  *rc = TSH_OK;
  MaskPart *p = m->parts[g].get();
  if (p->built_rows.load(std::memory_order_acquire) == s->rows) return p;
  // the shard grew since the part was built (appends take the shard exclusively, so no search that reads the old
  // buffers is running: every search that started after the append passes through here first)
  std::lock_guard<std::mutex> lk(m->mu);
  if (p->built_rows.load(std::memory_order_acquire) != s->rows) {
    *rc = mask_build_part(m, p, s);
    if (*rc) return nullptr;
  }
  return p;
}

}  // namespace

extern "C" {

int32_t tsh_abi_version(void) { return TSH_ABI_VERSION; }

int32_t tsh_device_count(void) { return device_count_cached(); }

int32_t tsh_last_error(char *buf, int32_t len) {
  if (buf && len > 0) {
    size_t n = std::min<size_t>(g_err.size(), (size_t)len - 1);
    memcpy(buf, g_err.data(), n);
    buf[n] = 0;
  }
  return (int32_t)g_err.size();
}

int32_t tsh_index_create(int32_t dim, int32_t metric, int64_t capacity_rows, int32_t n_devices,
                         tsh_index **out) {
  int rc = check_create_args(dim, metric, capacity_rows, out);
  if (rc) return rc;
  // TSH_SHARDS_SHARE_DEVICES=1 (testing hook, obeyed only after TSH_OPT_TEST_HOOKS): shards of a multi-device
  // handle may share a physical device (shard g on device g % count), so the multi-shard code path -- append
  // routing, one host thread per shard, host merge -- can be exercised on a one-GPU box
  const char *share_env = test_env("TSH_SHARDS_SHARE_DEVICES");
  const bool share = share_env && share_env[0] == '1';
  if (n_devices < 1 || (!share && n_devices > device_count_cached()) || n_devices > 64)
    return set_err(TSH_E_BAD_ARG, "n_devices %d outside [1,%d]", n_devices, device_count_cached());
  std::unique_ptr<tsh_index> idx(new tsh_index());
  idx->dim = dim;
  idx->metric = metric;
  if (n_devices == 1) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::unique_ptr<Shard> s;
    rc = make_shard(dim, metric, dev, 0, capacity_rows, &s);
    if (rc) return rc;
    idx->shards.push_back(std::move(s));
  } else {
    if (capacity_rows <= 0)
      return set_err(TSH_E_BAD_ARG, "multi-device index needs capacity_rows to place the row ranges");
    idx->rows_per_shard = round_up((capacity_rows + n_devices - 1) / n_devices, 64);
    for (int g = 0; g < n_devices; ++g) {
      std::unique_ptr<Shard> s;
      rc = make_shard(dim, metric, g % device_count_cached(), (int64_t)g * idx->rows_per_shard,
                      idx->rows_per_shard, &s);
      if (rc) {
        for (auto &p : idx->shards) shard_destroy(p.get());
        return rc;
      }
      idx->shards.push_back(std::move(s));
    }
    idx->workers.reset(new ShardWorkers(n_devices));
  }
  *out = idx.release();
  return TSH_OK;
}

int32_t tsh_index_create_shard(int32_t dim, int32_t metric, int64_t capacity_rows, int32_t device_id,
                               int64_t row_base, tsh_index **out) {
  int rc = check_create_args(dim, metric, capacity_rows, out);
  if (rc) return rc;
  if (row_base < 0) return set_err(TSH_E_BAD_ARG, "row_base < 0");
  int dev = device_id;
  if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (dev >= device_count_cached()) return set_err(TSH_E_BAD_ARG, "device %d not present", dev);
  std::unique_ptr<tsh_index> idx(new tsh_index());
  idx->dim = dim;
  idx->metric = metric;
  std::unique_ptr<Shard> s;
  rc = make_shard(dim, metric, dev, row_base, capacity_rows, &s);
  if (rc) return rc;
  idx->shards.push_back(std::move(s));
  *out = idx.release();
  return TSH_OK;
}

int32_t tsh_index_destroy(tsh_index *idx) {
  if (!idx) return TSH_OK;
  idx->workers.reset();  // joins the shard threads
  for (auto &s : idx->shards) {
    std::unique_lock<RwLock> xl(s->mu);
    shard_destroy(s.get());
  }
  delete idx;
  return TSH_OK;
}

int32_t tsh_index_append(tsh_index *idx, int64_t first_row_id, int64_t n_rows, const float *rows) {
  return index_append(idx, first_row_id, n_rows, rows, false);
}

int32_t tsh_index_append_device(tsh_index *idx, int64_t first_row_id, int64_t n_rows, const void *d_rows) {
  return index_append(idx, first_row_id, n_rows, static_cast<const float *>(d_rows), true);
}

int32_t tsh_index_set_deleted(tsh_index *idx, const int64_t *ids, int64_t n) {
  if (!idx) return set_err(TSH_E_BAD_ARG, "index is NULL");
  if (n < 0) return set_err(TSH_E_BAD_ARG, "n < 0");
  if (n == 0) return TSH_OK;
  if (!ids) return set_err(TSH_E_BAD_ARG, "ids is NULL");
  for (auto &sp : idx->shards) {
    Shard *s = sp.get();
    std::unique_lock<RwLock> xl(s->mu);
    if (s->rows == 0) continue;
    HIPCHK(hipSetDevice(s->device));
    if (n > s->del_ids_cap) {  // kept between calls (deletes arrive in small batches, journal flush by journal flush)
      hipFree(s->d_del_ids);
      s->d_del_ids = nullptr;
      s->del_ids_cap = 0;
      const int64_t want = std::max<int64_t>(round_up(n, 1024), 4096);
      HIPCHK(hipMalloc(&s->d_del_ids, (size_t)want * sizeof(int64_t)));
      s->del_ids_cap = want;
    }
    int64_t *d_ids = s->d_del_ids;
    hipStream_t st = s->ingest_stream;
    hipError_t e = hipMemcpyAsync(d_ids, ids, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(s->d_tmp_u32, 0, 4, st);
    if (e == hipSuccess) {
      int grid = (int)std::min<int64_t>((n + 255) / 256, 1024);
      live_clear_ids_kernel<<<grid, 256, 0, st>>>(s->d_live, d_ids, n, s->row_base, s->rows, s->d_tmp_u32);
    }
    uint32_t cleared = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&cleared, s->d_tmp_u32, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return set_err(TSH_E_HIP, "set_deleted: %s", hipGetErrorString(e));
    s->deleted += cleared;
    if (cleared) s->all_live = false;
    if (!s->quar_ids.empty()) {  // a quarantined row's live bit is clear already: take it off the list
      size_t before = s->quar_ids.size();
      for (int64_t i = 0; i < n; ++i) {
        const int64_t r = ids[i] - s->row_base;
        if (r < 0 || r >= s->rows) continue;
        auto it = std::lower_bound(s->quar_ids.begin(), s->quar_ids.end(), (uint32_t)r);
        if (it != s->quar_ids.end() && *it == (uint32_t)r) s->quar_ids.erase(it);
      }
      if (s->quar_ids.size() != before) {
        s->deleted += (int64_t)(before - s->quar_ids.size());
        int rc = quarantine_upload(s);
        if (rc) return rc;
      }
    }
  }
  return TSH_OK;
}

int64_t tsh_index_size(tsh_index *idx) {
  if (!idx) return 0;
  int64_t m = 0;
  for (auto &s : idx->shards) {
    std::shared_lock<RwLock> sl = share(idx, s.get());
    if (s->rows > 0) m = std::max(m, s->row_base + s->rows);
  }
  return m;
}
int32_t tsh_index_dim(tsh_index *idx) { return idx ? idx->dim : 0; }
int32_t tsh_index_metric(tsh_index *idx) { return idx ? idx->metric : 0; }

int64_t tsh_candidate_block_bytes(int32_t entries) {
  return (int64_t)sizeof(BlockHeader) + (int64_t)std::max(entries, 0) * (int64_t)sizeof(BlockEntry);
}
int32_t tsh_default_block_entries(int32_t k) {
  int64_t e = (int64_t)std::max(k, 1) + 156;
  e = round_up(e, 64);
  return (int32_t)std::min<int64_t>(e, 1 << 20);
}

static int32_t search_impl(tsh_index *idx, const float *queries, int32_t nq, int32_t k, double thr,
                           const uint8_t *row_mask, tsh_mask *mask_h, int64_t *out_ids, double *out_dist,
                           int32_t *out_count) {
  if (!idx) return set_err(TSH_E_BAD_ARG, "index is NULL");
  if (mask_h && mask_h->idx != idx) return set_err(TSH_E_BAD_ARG, "the mask handle was made for another index");
  if (nq < 0) return set_err(TSH_E_BAD_ARG, "nq < 0");
  if (nq == 0) return TSH_OK;
  if (!queries || !out_count) return set_err(TSH_E_BAD_ARG, "queries / out_count is NULL");
  for (int32_t q = 0; q < nq; ++q) out_count[q] = 0;
  if (k <= 0) return TSH_OK;  // reference: topK <= 0 yields an empty list
  if (!out_ids || !out_dist) return set_err(TSH_E_BAD_ARG, "out_ids / out_dist is NULL");

  const double t_in = now_us();
  size_t ns = idx->shards.size();
  int32_t entries = tsh_default_block_entries(k);
  size_t bb = (size_t)tsh_candidate_block_bytes(entries);
  std::vector<std::unique_ptr<uint8_t[]>> blocks(ns);  // uninitialised: every block is written whole
  std::vector<size_t> block_caps(ns, 0);
  struct GiveBack {
    tsh_index *idx;
    std::vector<std::unique_ptr<uint8_t[]>> *b;
    std::vector<size_t> *c;
    ~GiveBack() {
      for (size_t g = 0; g < b->size(); ++g)
        if ((*b)[g]) idx->give_blocks(std::move((*b)[g]), (*c)[g]);
    }
  } give_back{idx, &blocks, &block_caps};
  std::vector<std::vector<std::vector<BlockEntry>>> spills(ns), extras(ns);
  std::vector<int> rcs(ns, TSH_OK);
  std::vector<std::string> errs(ns);
  std::vector<char> active(ns, 0);
  std::vector<char> finalized((size_t)nq, 0);  // done early by the batched path's chunk callback

  auto run = [&](size_t g) {
    Shard *s = idx->shards[g].get();
    std::shared_lock<RwLock> sl = share(idx, s);
    if (s->rows == 0) return;
    active[g] = 1;
    blocks[g] = idx->take_blocks(bb * (size_t)nq, &block_caps[g]);
    spills[g].resize((size_t)nq);
    extras[g].resize((size_t)nq);
    SearchOut so;
    so.h_blocks = blocks[g].get();
    so.spill = &spills[g];
    so.extra = &extras[g];
    if (ns == 1)  // batched path: finalise a chunk of queries while the GPU still works on the next one
      so.on_chunk = [&](int32_t q0, int32_t q1, const char *skip, const uint8_t *base, const double *mag_a) {
        parallel_for_range(q0, q1, [&](int32_t q) {
          if (skip[q]) return;
          const uint8_t *b = base + (size_t)q * bb;
          const BlockHeader *h = reinterpret_cast<const BlockHeader *>(b);
          const std::vector<BlockEntry> &ex = extras[0][(size_t)q];
          const EntryList two[2] = {
              {reinterpret_cast<const BlockEntry *>(b + sizeof(BlockHeader)), std::min(h->count, h->entries)},
              {ex.data(), (uint32_t)ex.size()}};
          out_count[q] = finalize_query(idx->metric, idx->dim, queries + (size_t)q * idx->dim, k, thr, two,
                                        ex.empty() ? 1 : 2, out_ids + (size_t)q * k, out_dist + (size_t)q * k, mag_a + q);
          finalized[(size_t)q] = 1;
        });
      };
    if (ns == 1) {
      so.fin_thr = thr;
      so.on_final = [&](int32_t q0, int32_t q1, const char *skip, const int64_t *ids, const double *dist, const int32_t *cnt) {
        parallel_for_range(q0, q1, [&](int32_t q) {
          if (skip[q]) return;
          memcpy(out_ids + (size_t)q * k, ids + (size_t)q * k, (size_t)k * sizeof(int64_t));
          memcpy(out_dist + (size_t)q * k, dist + (size_t)q * k, (size_t)k * sizeof(double));
          out_count[q] = cnt[q];
          finalized[(size_t)q] = 1;
        });
      };
    }
    MaskSrc ms(row_mask);
    if (mask_h) {  // the handle's part for this shard: resident, rebuilt here if the shard grew since
      const MaskPart *mp = mask_part(mask_h, g, s, &rcs[g]);
      if (!mp) {
        errs[g] = g_err;
        return;
      }
      ms = MaskSrc(mp);
    }
    rcs[g] = shard_search_any(s, s->batch, idx->batch_min_nq.load(), queries, nq, k, ms, entries, &so);
    if (rcs[g]) errs[g] = g_err;
  };
  if (ns == 1) {
    run(0);
  } else {
    const std::function<void(size_t)> frun = run;
    if (!idx->workers || !idx->workers->run(frun)) {  // the workers are busy with a concurrent call: own threads
      std::vector<std::thread> th;
      for (size_t g = 1; g < ns; ++g) th.emplace_back(run, g);
      run(0);
      for (auto &t : th) t.join();
    }
  }
  for (size_t g = 0; g < ns; ++g)
    if (rcs[g]) {
      g_err = errs[g];
      return rcs[g];
    }
  const double t_shards = now_us();
  // (the batched path's chunk callback has usually done everything: waking the pool for nothing cost 30-50 us)
  const bool all_done = std::find(finalized.begin(), finalized.end(), (char)0) == finalized.end();
  if (!all_done) parallel_for(nq, [&](int32_t q) {
    if (finalized[(size_t)q]) return;
    std::vector<std::pair<const BlockEntry *, uint32_t>> lists;
    for (size_t g = 0; g < ns; ++g) {
      if (!active[g]) continue;
      if (!spills[g][(size_t)q].empty()) {
        lists.push_back({spills[g][(size_t)q].data(), (uint32_t)spills[g][(size_t)q].size()});
      } else {
        const uint8_t *b = blocks[g].get() + (size_t)q * bb;
        const BlockHeader *h = reinterpret_cast<const BlockHeader *>(b);
        lists.push_back({reinterpret_cast<const BlockEntry *>(b + sizeof(BlockHeader)),
                         std::min(h->count, h->entries)});
      }
      const std::vector<BlockEntry> &ex = extras[g][(size_t)q];
      if (!ex.empty()) lists.push_back({ex.data(), (uint32_t)ex.size()});
    }
    out_count[q] = finalize_query(idx->metric, idx->dim, queries + (size_t)q * idx->dim, k, thr, lists,
                                  out_ids + (size_t)q * k, out_dist + (size_t)q * k);
  });
  if (trace_batch() && nq >= 64)
    fprintf(stderr, "[tsh search] nq=%d shards %.0f us, finalize %.0f us\n", nq, t_shards - t_in, now_us() - t_shards);
  return TSH_OK;
}

int32_t tsh_search(tsh_index *idx, const float *queries, int32_t nq, int32_t k, double thr,
                   const uint8_t *row_mask, int64_t *out_ids, double *out_dist, int32_t *out_count) {
  return search_impl(idx, queries, nq, k, thr, row_mask, nullptr, out_ids, out_dist, out_count);
}

int32_t tsh_search_masked(tsh_index *idx, const float *queries, int32_t nq, int32_t k, double thr, tsh_mask *mask,
                          int64_t *out_ids, double *out_dist, int32_t *out_count) {
  return search_impl(idx, queries, nq, k, thr, nullptr, mask, out_ids, out_dist, out_count);
}

int32_t tsh_mask_create(tsh_index *idx, const uint8_t *bits, int64_t n_bytes, tsh_mask **out) {
  if (!out) return set_err(TSH_E_BAD_ARG, "out is NULL");
  *out = nullptr;
  if (!idx) return set_err(TSH_E_BAD_ARG, "index is NULL");
  if (n_bytes < 0 || (n_bytes > 0 && !bits)) return set_err(TSH_E_BAD_ARG, "bits is NULL / n_bytes < 0");
  std::unique_ptr<tsh_mask> m(new tsh_mask());
  m->idx = idx;
  try {
    m->bits.assign(bits, bits + n_bytes);
  } catch (...) {
    return set_err(TSH_E_OOM, "no memory for a %lld-byte mask", (long long)n_bytes);
  }
  for (size_t g = 0; g < idx->shards.size(); ++g) m->parts.emplace_back(new MaskPart());
  // every shard's part now: a handle exists to take this work out of the searches
  int rc = TSH_OK;
  {
    std::lock_guard<std::mutex> lk(m->mu);
    for (size_t g = 0; g < idx->shards.size() && rc == TSH_OK; ++g) {
      Shard *s = idx->shards[g].get();
      std::shared_lock<RwLock> sl = share(idx, s);
      rc = mask_build_part(m.get(), m->parts[g].get(), s);
    }
  }
  if (rc) {
    const std::string keep = g_err;
    for (auto &p : m->parts) mask_part_free(p.get());
    g_err = keep;
    return rc;
  }
  *out = m.release();
  return TSH_OK;
}

int32_t tsh_mask_destroy(tsh_mask *mask) {
  if (!mask) return TSH_OK;
  for (auto &p : mask->parts) mask_part_free(p.get());
  delete mask;
  return TSH_OK;
}

int64_t tsh_mask_kept(tsh_mask *mask) {
  if (!mask) return set_err(TSH_E_BAD_ARG, "mask is NULL");
  tsh_index *idx = mask->idx;
  int64_t kept = 0;
  for (size_t g = 0; g < idx->shards.size(); ++g) {
    Shard *s = idx->shards[g].get();
    std::shared_lock<RwLock> sl = share(idx, s);
    int rc = TSH_OK;
    const MaskPart *p = mask_part(mask, g, s, &rc);
    if (!p) return rc;
    kept += p->kept;
  }
  return kept;
}

// ---- asynchronous single-query searches ------------------------------------------
int32_t tsh_max_inflight(void) { return MAX_CTX; }

static int32_t submit_impl(tsh_index *idx, const float *query, int32_t k, const uint8_t *row_mask, tsh_mask *mask_h,
                           int32_t *out_ticket) {
  if (!idx || !query || !out_ticket) return set_err(TSH_E_BAD_ARG, "NULL pointer");
  if (k <= 0) return set_err(TSH_E_BAD_ARG, "k <= 0");
  if (mask_h && mask_h->idx != idx) return set_err(TSH_E_BAD_ARG, "the mask handle was made for another index");
  *out_ticket = -1;
  std::unique_ptr<Ticket> t(new Ticket());
  t->k = k;
  t->entries = tsh_default_block_entries(k);
  t->query.assign(query, query + idx->dim);
  t->jobs.resize(idx->shards.size());
  t->locks.resize(idx->shards.size());
  int rc = TSH_OK;
  // The handle stays share-locked from submit to wait.  A caller that already has tickets open must not queue
  // behind a waiting append / delete (it would never reach the wait that lets the writer in): it is told to
  // drain its tickets first, which also keeps a caller that always has one open from starving the writer.
  const bool holding = idx->tickets_open.load() > 0;
  if (holding)
    for (auto &sp : idx->shards)
      if (sp->mu.writer_pending())
        return set_err(TSH_E_BUSY, "an append / delete is waiting for this handle's open asynchronous searches: "
                                   "wait for them, then submit again");
  for (size_t g = 0; g < idx->shards.size() && rc == TSH_OK; ++g) {
    Shard *s = idx->shards[g].get();
    s->mu.lock_shared_gate(holding);
    t->locks[g] = std::shared_lock<RwLock>(s->mu, std::adopt_lock);
    if (s->rows == 0) continue;
    // Contexts held by synchronous callers come back on their own, so wait for one -- unless the
    // un-waited tickets alone could hold them all: waiting would then never end for a caller that
    // submits before it waits
    Ctx *c = ctx_acquire(s, idx->tickets_open.load() < MAX_CTX);
    if (!c) {
      rc = set_err(TSH_E_BUSY, "%d asynchronous searches are already in flight on this handle", MAX_CTX);
      break;
    }
    t->jobs[g].c = c;
    std::vector<uint64_t> words;
    std::vector<uint32_t> list_ids;  // (job_enqueue copies what it keeps of either before it returns)
    RowList list;
    int64_t rows_est = 0;
    uint64_t epoch = 0;
    const MaskPart *mp = nullptr;
    if (mask_h) {
      mp = mask_part(mask_h, g, s, &rc);
      if (!mp) break;
      rows_est = std::max<int64_t>(mp->kept, 1);
      if (mp->list_padded > 0 && row_list_pays(s, mp->kept, k, t->entries)) {
        list.d_ids = mp->d_list;
        list.padded = mp->list_padded;
      }
    } else if (row_mask) {
      int32_t n_tiles = (int32_t)((s->rows + 63) / 64);
      words.resize((size_t)n_tiles);
      slice_mask(s, row_mask, words.data(), n_tiles);
      epoch = s->mask_epoch_src.fetch_add(1);
      rows_est = popcount_words(words.data(), words.size());
      if (rows_est == 0) rows_est = 1;
      if (build_row_list(s, words.data(), n_tiles, rows_est, k, t->entries, &list_ids)) {  // a selective mask: its rows as a list
        list.ids = list_ids.data();
        list.padded = (int32_t)list_ids.size();
      }
    }
    rc = job_enqueue(s, &t->jobs[g], query, k, t->entries, mp ? mp->h_words.data() : (row_mask ? words.data() : nullptr),
                     epoch, nullptr, rows_est, (list.ids || list.d_ids) ? &list : nullptr, false, false, 0, mp);
  }
  if (rc != TSH_OK) {
    std::string keep = g_err;
    for (size_t g = 0; g < idx->shards.size(); ++g)
      if (t->jobs[g].c) {
        if (t->jobs[g].counted) hipEventSynchronize(t->jobs[g].c->ev_done);
        if (t->jobs[g].counted) idx->shards[g]->inflight.fetch_sub(1);
        ctx_release(idx->shards[g].get(), t->jobs[g].c);
      }
    g_err = keep;
    return rc;
  }
  std::lock_guard<std::mutex> lk(idx->tk_mu);
  int slot = -1;
  for (size_t i = 0; i < idx->tickets.size(); ++i)
    if (!idx->tickets[i]) {
      slot = (int)i;
      break;
    }
  if (slot < 0) {
    idx->tickets.emplace_back();
    slot = (int)idx->tickets.size() - 1;
  }
  idx->tickets[(size_t)slot] = std::move(t);
  idx->tickets_open.fetch_add(1);
  *out_ticket = slot;
  return TSH_OK;
}

int32_t tsh_search_submit(tsh_index *idx, const float *query, int32_t k, const uint8_t *row_mask, int32_t *out_ticket) {
  return submit_impl(idx, query, k, row_mask, nullptr, out_ticket);
}
int32_t tsh_search_submit_masked(tsh_index *idx, const float *query, int32_t k, tsh_mask *mask, int32_t *out_ticket) {
  return submit_impl(idx, query, k, nullptr, mask, out_ticket);
}

int32_t tsh_search_ready(tsh_index *idx, int32_t ticket) {
  if (!idx) return set_err(TSH_E_BAD_ARG, "index is NULL");
  std::lock_guard<std::mutex> lk(idx->tk_mu);
  if (ticket < 0 || (size_t)ticket >= idx->tickets.size() || !idx->tickets[(size_t)ticket])
    return set_err(TSH_E_BAD_ARG, "unknown ticket %d", ticket);
  Ticket *t = idx->tickets[(size_t)ticket].get();
  for (size_t g = 0; g < idx->shards.size(); ++g) {
    if (!t->jobs[g].c) continue;
    if (hipSetDevice(idx->shards[g]->device) != hipSuccess) return set_err(TSH_E_HIP, "hipSetDevice failed");
    const hipError_t e = hipEventQuery(t->jobs[g].c->ev_done);
    if (e == hipErrorNotReady) return 0;
    if (e != hipSuccess) return set_err(TSH_E_HIP, "hipEventQuery failed: %s", hipGetErrorString(e));
  }
  return 1;
}

int32_t tsh_search_wait(tsh_index *idx, int32_t ticket, double thr, int64_t *out_ids, double *out_dist,
                        int32_t *out_count) {
  if (!idx || !out_ids || !out_dist || !out_count) return set_err(TSH_E_BAD_ARG, "NULL pointer");
  std::unique_ptr<Ticket> t;
  {
    std::lock_guard<std::mutex> lk(idx->tk_mu);
    if (ticket < 0 || (size_t)ticket >= idx->tickets.size() || !idx->tickets[(size_t)ticket])
      return set_err(TSH_E_BAD_ARG, "unknown ticket %d", ticket);
    t = std::move(idx->tickets[(size_t)ticket]);
    idx->tickets_open.fetch_sub(1);
  }
  *out_count = 0;
  int rc = TSH_OK;
  std::vector<std::vector<BlockEntry>> spills(idx->shards.size()), extras(idx->shards.size());
  std::vector<std::pair<const BlockEntry *, uint32_t>> lists;
  for (size_t g = 0; g < idx->shards.size(); ++g) {
    Job &j = t->jobs[g];
    if (!j.c) continue;
    Shard *s = idx->shards[g].get();
    if (rc == TSH_OK) {
      rc = job_finish(s, &j, &spills[g], &extras[g]);
    } else {
      if (j.counted) hipEventSynchronize(j.c->ev_done);
      if (j.counted) s->inflight.fetch_sub(1);
      j.counted = false;
    }
    if (rc == TSH_OK) {
      if (!spills[g].empty()) {
        lists.push_back({spills[g].data(), (uint32_t)spills[g].size()});
      } else {
        const BlockHeader *h = reinterpret_cast<const BlockHeader *>(j.c->h_block);
        lists.push_back({reinterpret_cast<const BlockEntry *>(j.c->h_block + sizeof(BlockHeader)),
                         std::min(h->count, h->entries)});
      }
      if (!extras[g].empty()) lists.push_back({extras[g].data(), (uint32_t)extras[g].size()});
    }
  }
  if (rc == TSH_OK)
    *out_count = finalize_query(idx->metric, idx->dim, t->query.data(), t->k, thr, lists, out_ids, out_dist);
  std::string keep = g_err;
  for (size_t g = 0; g < idx->shards.size(); ++g)
    if (t->jobs[g].c) ctx_release(idx->shards[g].get(), t->jobs[g].c);
  g_err = keep;
  return rc;
}

int32_t tsh_search_shard(tsh_index *idx, const float *queries, int32_t nq, int32_t k,
                         const uint8_t *row_mask, int32_t entries, void *d_out_blocks, void *stream) {
  if (!idx || idx->shards.size() != 1) return set_err(TSH_E_BAD_ARG, "needs a single-shard handle");
  if (nq <= 0 || !queries || !d_out_blocks || k <= 0 || entries < 1)
    return set_err(TSH_E_BAD_ARG, "bad nq / k / entries / pointers");
  Shard *s = idx->shards[0].get();
  std::shared_lock<RwLock> sl = share(idx, s);
  size_t bb = (size_t)tsh_candidate_block_bytes(entries);
  if (s->rows == 0) {  // an empty shard contributes empty blocks
    HIPCHK(hipSetDevice(s->device));
    std::vector<uint8_t> z(bb * (size_t)nq, 0);
    for (int32_t q = 0; q < nq; ++q) {
      BlockHeader *h = reinterpret_cast<BlockHeader *>(z.data() + (size_t)q * bb);
      h->entries = (uint32_t)entries;
      h->k = (uint32_t)k;
      h->metric = (uint32_t)s->metric;
      h->row_base = s->row_base;
    }
    HIPCHK(hipMemcpy(d_out_blocks, z.data(), z.size(), hipMemcpyHostToDevice));
    return TSH_OK;
  }
  SearchOut so;
  so.d_blocks = static_cast<uint8_t *>(d_out_blocks);
  so.user_stream = static_cast<hipStream_t>(stream);
  return shard_search_any(s, s->batch, idx->batch_min_nq.load(), queries, nq, k, row_mask, entries, &so);
}

// ---- progressive shard search ---------------------------------------------------------------------------------
// tsh_search_shard answers when the LAST of its queries is done; a caller that exchanges the blocks group by group
// (tsh_search_sharded, sharded.py) then starts group g + 1's scans only after group g's have drained, and every
// group pays the fill and the drain of the scan pipeline (measured: ~45 us per group on a 125 k-row shard, a fifth of
// a ten-query group).  Here the scans of ALL nq queries run as ONE pipeline on a library thread -- group boundaries
// do not exist for the GPU -- and the caller is told how many LEADING queries' blocks are final.
struct tsh_shard_stream {
  tsh_index *idx = nullptr;
  const float *queries = nullptr;
  const uint8_t *mask = nullptr;
  std::vector<float> own_queries;  // public entry points: the caller's arrays are consumed before begin returns
  std::vector<uint8_t> own_mask;
  int32_t nq = 0, k = 0, entries = 0, step = 0;
  uint8_t *d_blocks = nullptr;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<char> done_q;  // single-query route: query q's block is final
  int32_t done = 0;          // leading queries whose blocks are final
  bool finished = false;     // the worker has returned
  std::atomic<int32_t> done_seen{0};  // copies of the two for a caller that polls before it sleeps
  std::atomic<bool> finished_seen{false};
  // single-query route: leading queries whose kernels are all enqueued, and the event behind each one's block
  // writer -- what a caller needs to order its own stream behind a group's blocks without waiting for them
  std::vector<char> enq_q;
  std::vector<hipEvent_t> ev_q;
  std::vector<hipStream_t> st_q;  // the stream ev_q[q] was recorded on
  std::vector<uint64_t> seq_q;    // ... and when (the device's enqueue order)
  int32_t enq = 0;
  std::atomic<int32_t> enq_seen{0};
  std::atomic<int> route{0};  // 0 not decided yet, 1 every query its own scan (events exist), 2 anything else
  uint32_t tag = 0;           // generation the blocks carry (BlockHeader.pad[1])
  bool leave_overflow = false;  // the caller enqueues its exchange ahead of the blocks (Job::leave_overflow)
  int rc = TSH_OK;
  std::string err;
  double busy_us = 0;  // worker: first enqueue to last block
  std::thread th;              // the search's own thread ...
  OneWorker *exec = nullptr;   // ... or the caller's persistent one (tsh_search_sharded: starting a thread costs a
                               // 20-query call on a 125 k-row shard 30-40 us before its first scan is enqueued)

  void publish(int32_t upto) {  // queries [0, upto) are final
    {
      std::lock_guard<std::mutex> lk(mu);
      done = std::max(done, upto);
      done_seen.store(done, std::memory_order_release);
    }
    cv.notify_all();
  }
  void mark_enqueued(int32_t q, hipEvent_t ev, hipStream_t where, uint64_t seq) {
    bool moved = false;
    {
      std::lock_guard<std::mutex> lk(mu);
      ev_q[(size_t)q] = ev;
      st_q[(size_t)q] = where;
      seq_q[(size_t)q] = seq;
      enq_q[(size_t)q] = 1;
      while (enq < nq && enq_q[(size_t)enq]) {
        ++enq;
        moved = true;
      }
      enq_seen.store(enq, std::memory_order_release);
    }
    if (moved) cv.notify_all();
  }
  void mark(int32_t q) {
    bool moved = false;
    {
      std::lock_guard<std::mutex> lk(mu);
      done_q[(size_t)q] = 1;
      while (done < nq && done_q[(size_t)done]) {
        ++done;
        moved = true;
      }
      done_seen.store(done, std::memory_order_release);
    }
    if (moved) cv.notify_all();
  }
  int body() {
    Shard *s = idx->shards[0].get();
    std::shared_lock<RwLock> sl = share(idx, s);
    HIPCHK(hipSetDevice(s->device));
    if (!own_mask.empty()) {
      // _begin copied the mask for the rows the shard had THEN; rows appended between that look and this lock are
      // beyond the caller's mask: not kept (zero bits), and never read past the copy's end
      const size_t need = (size_t)((s->row_base + s->rows + 7) / 8);
      if (own_mask.size() < need) {
        own_mask.resize(need, 0);
        mask = own_mask.data();
      }
    }
    const size_t bb = (size_t)tsh_candidate_block_bytes(entries);
    if (s->rows == 0) {  // an empty shard contributes empty blocks
      std::vector<uint8_t> z(bb * (size_t)nq, 0);
      for (int32_t q = 0; q < nq; ++q) {
        BlockHeader *h = reinterpret_cast<BlockHeader *>(z.data() + (size_t)q * bb);
        h->entries = (uint32_t)entries;
        h->k = (uint32_t)k;
        h->metric = (uint32_t)s->metric;
        h->row_base = s->row_base;
        h->pad[1] = tag;
      }
      route.store(2, std::memory_order_release);
      HIPCHK(hipMemcpy(d_blocks, z.data(), z.size(), hipMemcpyHostToDevice));
      publish(nq);
      return TSH_OK;
    }
    const int32_t min_nq = idx->batch_min_nq.load();
    const int32_t st = step > 0 ? std::min(step, nq) : nq;
    if (!shard_takes_batch(s, min_nq, st, k)) {  // one pipeline over all queries
      done_q.assign((size_t)nq, 0);
      enq_q.assign((size_t)nq, 0);
      ev_q.assign((size_t)nq, nullptr);
      st_q.assign((size_t)nq, nullptr);
      seq_q.assign((size_t)nq, 0);
      SearchOut so;
      so.d_blocks = d_blocks;
      so.tag = tag;
      so.leave_overflow = leave_overflow;
      so.on_done = [this](int32_t q) { mark(q); };
      so.on_enqueued = [this](int32_t q, hipEvent_t ev, hipStream_t where, uint64_t seq) { mark_enqueued(q, ev, where, seq); };
      route.store(1, std::memory_order_release);
      cv.notify_all();
      return shard_search_blocks(s, queries, nq, k, mask, entries, &so, PIPE_DEPTH);
    }
    route.store(2, std::memory_order_release);
    cv.notify_all();
    // matrix-core route: a call per step (the batched path answers a call as a whole; the next step's call overlaps
    // the caller's work on this one as before)
    for (int32_t q0 = 0; q0 < nq; q0 += st) {
      const int32_t gq = std::min(st, nq - q0);
      SearchOut so;
      so.d_blocks = d_blocks + (size_t)q0 * bb;
      so.tag = tag;  // (queries the batch hands back to the single-query path carry it at once)
      int r = shard_search_any(s, s->batch, min_nq, queries + (size_t)q0 * s->dim, gq, k, mask, entries, &so);
      if (r) return r;
      if (tag) {  // the matrix-core path's blocks get their generation now (host-synchronised: they are final)
        stamp_tag_kernel<<<(unsigned)((gq + 63) / 64), 64, 0, s->aux_stream>>>(so.d_blocks, bb, gq, tag);
        HIPCHK(hipStreamSynchronize(s->aux_stream));
      }
      publish(q0 + gq);
    }
    return TSH_OK;
  }
  void run() {
    const double t0 = now_us();
    int r = body();
    std::string e = r ? g_err : std::string();
    {
      std::lock_guard<std::mutex> lk(mu);
      rc = r;
      err = e;
      finished = true;
      busy_us = now_us() - t0;
      if (route.load() == 0) route.store(2);
      finished_seen.store(true, std::memory_order_release);
    }
    cv.notify_all();
  }
};

namespace {
int shard_stream_begin(tsh_index *idx, const float *queries, int32_t nq, int32_t k, const uint8_t *row_mask,
                       int32_t entries, void *d_out_blocks, int32_t step, bool copy_inputs, tsh_shard_stream **out,
                       uint32_t tag = 0, OneWorker *exec = nullptr, bool leave_overflow = false) {
  if (!out) return set_err(TSH_E_BAD_ARG, "out is NULL");
  *out = nullptr;
  if (!idx || idx->shards.size() != 1) return set_err(TSH_E_BAD_ARG, "needs a single-shard handle");
  if (nq <= 0 || !queries || !d_out_blocks || k <= 0 || entries < 1 || step < 0)
    return set_err(TSH_E_BAD_ARG, "bad nq / k / entries / step / pointers");
  std::unique_ptr<tsh_shard_stream> st(new tsh_shard_stream());
  st->idx = idx;
  st->nq = nq;
  st->k = k;
  st->entries = entries;
  st->step = step;
  st->tag = tag;
  st->leave_overflow = leave_overflow;
  st->d_blocks = static_cast<uint8_t *>(d_out_blocks);
  st->queries = queries;
  st->mask = row_mask;
  if (copy_inputs) {
    Shard *s = idx->shards[0].get();
    st->own_queries.assign(queries, queries + (size_t)nq * (size_t)s->dim);
    st->queries = st->own_queries.data();
    if (row_mask) {
      int64_t bits;
      {
        std::shared_lock<RwLock> sl = share(idx, s);
        bits = s->row_base + s->rows;  // the mask is global: one bit per row id below this shard's end
      }
      st->own_mask.assign(row_mask, row_mask + (size_t)((bits + 7) / 8));
      st->mask = st->own_mask.data();
    }
  }
  tsh_shard_stream *p = st.get();
  if (exec) {
    st->exec = exec;
    exec->post([p] { p->run(); });
  } else {
    try {
      st->th = std::thread([p] { p->run(); });
    } catch (...) {
      return set_err(TSH_E_OOM, "could not start the search thread");
    }
  }
  *out = st.release();
  return TSH_OK;
}

// Blocks until the kernels of the first min(want, nq) queries are all enqueued -- or it is clear they never will be
// one by one (matrix-core route, an empty shard, a failure).  -> the number of events written to `after` (at most
// `max`): the events behind the block writers of those queries that may still be running (a job further back than
// the pipeline is deep has been finished by the host); 0 = no stream order to be had, wait for _progress instead.
// (An event may meanwhile belong to a LATER query of the same pipeline -- its context was reused: waiting for it
// then waits a little longer than needed, never too short.)
int shard_stream_enqueued(tsh_shard_stream *st, int32_t want, hipEvent_t *after, int max) {
  want = std::min(want, st->nq);
  if (!blocking_wait()) {
    const double t_end = now_us() + 3000.0;
    while (st->enq_seen.load(std::memory_order_acquire) < want && st->route.load(std::memory_order_acquire) != 2 &&
           !st->finished_seen.load(std::memory_order_acquire) && now_us() < t_end)
      cpu_relax();
  }
  std::unique_lock<std::mutex> lk(st->mu);
  st->cv.wait(lk, [&] { return st->enq >= want || st->route.load() == 2 || st->finished; });
  if (st->route.load() != 1 || st->enq < want) return 0;
  // of the jobs that share a stream only the one enqueued LAST matters (a stream's kernels finish in order), so this
  // is two or three events, not eight: every wait is a packet the consumer's queue has to work through
  int n = 0;
  hipStream_t seen[MAX_CTX];
  uint64_t seen_seq[MAX_CTX];
  for (int32_t q = want - 1; q >= std::max(0, want - (int32_t)MAX_CTX); --q) {
    if (!st->ev_q[(size_t)q] || st->done_q[(size_t)q]) continue;
    int at = -1;
    for (int i = 0; i < n; ++i)
      if (seen[i] == st->st_q[(size_t)q]) at = i;
    if (at < 0) {
      if (n == max) continue;  // (cannot happen: max >= MAX_CTX jobs)
      at = n++;
      seen[at] = st->st_q[(size_t)q];
      seen_seq[at] = 0;
    }
    if (st->seq_q[(size_t)q] >= seen_seq[at]) {
      seen_seq[at] = st->seq_q[(size_t)q];
      after[at] = st->ev_q[(size_t)q];
    }
  }
  if (n == 0) after[n++] = st->ev_q[(size_t)(want - 1)];  // (all final already: any completed event will do)
  return n;
}

// blocks until min(want, nq) leading queries are final, or the search has ended; -> its status so far
int shard_stream_progress(tsh_shard_stream *st, int32_t want, int32_t *out_done) {
  want = std::min(want, st->nq);
  if (!blocking_wait()) {
    // the caller has nothing else to do and (blocking_wait: at least three CPUs per rank) a core to do it on: poll for
    // a few milliseconds before sleeping -- being woken through the condition variable costs 20-50 us, on the last
    // group of a call in full view
    const double t_end = now_us() + 3000.0;
    while (st->done_seen.load(std::memory_order_acquire) < want && !st->finished_seen.load(std::memory_order_acquire) &&
           now_us() < t_end)
      cpu_relax();
  }
  std::unique_lock<std::mutex> lk(st->mu);
  st->cv.wait(lk, [&] { return st->done >= want || st->finished; });
  if (out_done) *out_done = st->done;
  if (st->done >= want) return TSH_OK;  // (whatever happens to later queries)
  g_err = st->err;  // the search ended before it got there
  return st->rc != TSH_OK ? st->rc : set_err(TSH_E_HIP, "the shard search ended early");
}

int shard_stream_end(tsh_shard_stream *st, double *busy_us = nullptr) {
  if (st->exec) st->exec->wait();
  else if (st->th.joinable()) st->th.join();
  const int rc = st->rc;
  if (rc) g_err = st->err;
  if (busy_us) *busy_us = st->busy_us;
  delete st;
  return rc;
}
}  // namespace

int32_t tsh_search_shard_begin(tsh_index *idx, const float *queries, int32_t nq, int32_t k, const uint8_t *row_mask,
                               int32_t entries, void *d_out_blocks, int32_t step, tsh_shard_stream **out) {
  return shard_stream_begin(idx, queries, nq, k, row_mask, entries, d_out_blocks, step, /*copy_inputs=*/true, out);
}
int32_t tsh_search_shard_progress(tsh_shard_stream *st, int32_t want, int32_t *out_done) {
  if (!st) return set_err(TSH_E_BAD_ARG, "stream is NULL");
  return shard_stream_progress(st, want, out_done);
}
int32_t tsh_search_shard_end(tsh_shard_stream *st) {
  if (!st) return TSH_OK;
  return shard_stream_end(st);
}

int32_t tsh_merge_candidates(int32_t metric, int32_t dim, const float *queries, int32_t nq, int32_t k,
                             double thr, const void *blocks, int32_t n_blocks, int32_t entries,
                             int64_t *out_ids, double *out_dist, int32_t *out_count,
                             int32_t *needed_entries) {
  if (metric < 0 || metric > 2 || dim <= 0 || nq < 0 || n_blocks < 0 || entries < 0)
    return set_err(TSH_E_BAD_ARG, "bad metric / dim / nq / n_blocks / entries");
  if (nq == 0) return TSH_OK;
  if (!queries || !blocks || !out_count) return set_err(TSH_E_BAD_ARG, "NULL pointer");
  for (int32_t q = 0; q < nq; ++q) out_count[q] = 0;
  if (needed_entries) *needed_entries = entries;
  if (k <= 0) return TSH_OK;
  if (!out_ids || !out_dist) return set_err(TSH_E_BAD_ARG, "out_ids / out_dist is NULL");
  size_t bb = (size_t)tsh_candidate_block_bytes(entries);
  const uint8_t *base = static_cast<const uint8_t *>(blocks);
  uint32_t need = 0;
  for (int32_t b = 0; b < n_blocks; ++b)
    for (int32_t q = 0; q < nq; ++q) {
      const BlockHeader *h = reinterpret_cast<const BlockHeader *>(base + ((size_t)b * nq + q) * bb);
      if (h->entries != (uint32_t)entries)
        return set_err(TSH_E_FORMAT, "block %d/%d entries %u != %d", b, q, h->entries, entries);
      if (h->count > h->entries) need = std::max(need, h->count);
    }
  if (need) {
    if (needed_entries) *needed_entries = (int32_t)round_up(need, 64);
    return set_err(TSH_E_OVERFLOW, "a candidate block needs %u entries (have %d)", need, entries);
  }
  parallel_for(nq, [&](int32_t q) {
    std::vector<std::pair<const BlockEntry *, uint32_t>> lists;
    for (int32_t b = 0; b < n_blocks; ++b) {
      const uint8_t *p = base + ((size_t)b * nq + q) * bb;
      const BlockHeader *h = reinterpret_cast<const BlockHeader *>(p);
      lists.push_back({reinterpret_cast<const BlockEntry *>(p + sizeof(BlockHeader)), h->count});
    }
    out_count[q] = finalize_query(metric, dim, queries + (size_t)q * dim, k, thr, lists,
                                  out_ids + (size_t)q * k, out_dist + (size_t)q * k);
  });
  return TSH_OK;
}

int32_t tsh_get_counters(tsh_index *idx, tsh_counters *out) {
  if (!idx || !out) return set_err(TSH_E_BAD_ARG, "NULL pointer");
  memset(out, 0, sizeof *out);
  for (auto &sp : idx->shards) {
    Shard *s = sp.get();
    std::shared_lock<RwLock> sl = share(idx, s);
    if (s->rows > 0) out->rows = std::max(out->rows, s->row_base + s->rows);
    out->deleted_rows += s->deleted;
    out->searches = std::max<int64_t>(out->searches, s->c_searches.load());
    out->scan_launches += s->c_scans.load();
    out->batch_launches += s->c_batches.load();
    out->fallback_searches += s->c_fallbacks.load();
    out->batch_plane_fallbacks += s->c_plane_fallbacks.load();
    out->batch_scan_fallbacks += s->c_scan_fallbacks.load();
    out->list_scans += s->c_list_scans.load();
    out->exact_scans += s->c_exact_scans.load();
    out->exact_redone += s->c_pick_redone.load();
    out->candidates_total += s->c_cands.load();
    int64_t b = s->bytes;
    {
      std::lock_guard<std::mutex> lk(s->ctx_mu);
      for (auto &c : s->ctx_all) b += c->bytes;
      out->scan_us_sum += s->scan_us_sum;
      out->scan_us_samples += s->scan_us_samples;
      out->batch_kernel_last = s->batch_kernel_last;
    }
    out->bytes_resident += b;
    if (s->safe_mode()) out->safe_mode = 1;
    out->quarantined_rows += (int32_t)s->quar_ids.size();
    out->device_id = s->device;
  }
  return TSH_OK;
}

int32_t tsh_bench_scan(tsh_index *idx, const float *query, int32_t iters, const uint8_t *row_mask,
                       double *out_avg_us) {
  if (!idx || idx->shards.size() != 1 || !query || iters <= 0 || !out_avg_us)
    return set_err(TSH_E_BAD_ARG, "bad arguments");
  Shard *s = idx->shards[0].get();
  std::shared_lock<RwLock> sl = share(idx, s);
  if (s->rows == 0) return set_err(TSH_E_BAD_ARG, "empty index");
  Ctx *c = ctx_acquire(s, true);
  struct Rel {
    Shard *s;
    Ctx *c;
    ~Rel() { ctx_release(s, c); }
  } rel{s, c};
  int rc = ctx_prepare(s, c, tsh_default_block_entries(100), row_mask != nullptr);
  if (rc) return rc;
  hipStream_t st = s->aux_stream;
  bool masked = row_mask != nullptr || !s->all_live;
  int32_t n_tiles = (int32_t)((s->rows + 63) / 64);
  if (row_mask) {
    slice_mask(s, row_mask, c->h_mask, n_tiles);
    HIPCHK(hipMemcpyAsync(c->d_mask, c->h_mask, (size_t)n_tiles * 8, hipMemcpyHostToDevice, st));
    c->mask_epoch = 0;
  }
  memcpy(c->h_query, query, (size_t)s->dim * sizeof(float));
  for (int64_t j = s->dim; j < s->ld; ++j) c->h_query[j] = 0.f;
  HIPCHK(hipMemcpyAsync(c->d_query, c->h_query, (size_t)s->ld * sizeof(float), hipMemcpyHostToDevice, st));
  static thread_local ScanArgsQ sa;
  fill_scan_args(s, c, masked, row_mask != nullptr, &sa);
  int64_t live_rows = s->rows - s->deleted;
  if (row_mask) {
    live_rows = 0;
    live_rows = popcount_words(c->h_mask, (size_t)n_tiles);
  }
  const bool ml = masked && scan_mostly_live(live_rows, s->rows);
  std::vector<uint32_t> list_ids;  // the kernel a search with this mask would run: the list scan for selective ones
  const bool use_list = row_mask && build_row_list(s, c->h_mask, n_tiles, live_rows, 100, tsh_default_block_entries(100), &list_ids);
  if (use_list) {
    if ((rc = ctx_reserve_list(c, (int64_t)list_ids.size()))) return rc;
    memcpy(c->h_list, list_ids.data(), list_ids.size() * sizeof(uint32_t));
    c->list_epoch = 0;
    HIPCHK(hipMemcpyAsync(c->d_list, c->h_list, list_ids.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    sa.a.list = c->d_list;
    sa.a.n_tiles = (int32_t)(list_ids.size() / 64);
  }
  // (a search with few enough rows to look at takes their exact sums instead: that kernel, then)
  const int64_t n_exam = use_list ? (int64_t)list_ids.size() : s->rows;
  const bool exact = exact_applies(s, n_exam, 100, tsh_default_block_entries(100));
  static thread_local ExactArgsQ xa;
  if (exact) {
    if ((rc = ctx_reserve_exact(c, n_exam))) return rc;
    fill_exact_args(s, c, use_list, row_mask && !use_list, n_exam, c->h_query, &xa);
  }
  auto launch = [&]() {
    if (exact) launch_exact_scan(xa, s->metric, st, LaunchEv());
    else if (use_list) launch_scan_list(sa, s->nch, s->metric, st);
    else launch_scan(sa, s->nch, s->metric, masked, st, LaunchEv(), ml);
  };
  launch();  // warm
  HIPCHK(hipEventRecord(c->ev0, st));
  for (int32_t i = 0; i < iters; ++i) launch();
  HIPCHK(hipEventRecord(c->ev1, st));
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  *out_avg_us = (double)ms * 1000.0 / iters;
  return TSH_OK;
}

int32_t tsh_index_set_option(tsh_index *idx, int32_t option, int64_t value) {
  if (option == TSH_OPT_EXCHANGE_AHEAD) {  // process-wide (idx is ignored)
    if (value != 0 && value != 1) return set_err(TSH_E_BAD_ARG, "exchange ahead: 0 or 1");
    exchange_ahead_flag().store(value != 0, std::memory_order_release);
    return TSH_OK;
  }
  if (option == TSH_OPT_TEST_HOOKS) {  // process-wide (idx is ignored): see test_env, tsh_host_sync.h
    if (value != 0 && value != TSH_TEST_HOOKS_MAGIC) return set_err(TSH_E_BAD_ARG, "test hooks: wrong magic");
    test_hooks_flag().store(value != 0, std::memory_order_release);
    return TSH_OK;
  }
  if (!idx) return set_err(TSH_E_BAD_ARG, "index is NULL");
  if (option == TSH_OPT_BATCH_MIN_NQ) {
    if (value < 0 || value > (1 << 20)) return set_err(TSH_E_BAD_ARG, "batch_min_nq out of range");
    idx->batch_min_nq = (int32_t)value;
    return TSH_OK;
  }
  if (option == TSH_OPT_EXACT_SCAN_ROWS) {
    if (value < 0 || value > EX_MAX_ROWS) return set_err(TSH_E_BAD_ARG, "exact scan rows: 0 .. %d", EX_MAX_ROWS);
    for (auto &sh : idx->shards) {
      std::unique_lock<RwLock> xl(sh->mu);
      sh->exact_rows = (int)value;
    }
    return TSH_OK;
  }
  if (option == TSH_OPT_BATCH_GROUP) {
    if (value != 0 && value != 1) return set_err(TSH_E_BAD_ARG, "batch group: 0 or 1");
    for (auto &sh : idx->shards) {
      std::unique_lock<RwLock> xl(sh->mu);
      sh->batch_group = value != 0;
    }
    return TSH_OK;
  }
  if (option == TSH_OPT_BATCH_HUB) {
    if (value != 0 && value != 1) return set_err(TSH_E_BAD_ARG, "batch hub: 0 or 1");
    for (auto &sh : idx->shards) {
      std::unique_lock<RwLock> xl(sh->mu);
      sh->batch_hub = value != 0;
    }
    return TSH_OK;
  }
  if (option == TSH_OPT_EXACT_SELECT) {
    if (value != 0 && value != 1) return set_err(TSH_E_BAD_ARG, "exact select: 0 (one workgroup ranks k rows) or 1 (wide pick)");
    for (auto &sh : idx->shards) {
      std::unique_lock<RwLock> xl(sh->mu);
      sh->exact_pick = value != 0;
    }
    return TSH_OK;
  }
  if (option == TSH_OPT_BATCH_KERNEL) {
    if (value < 0 || value > 3) return set_err(TSH_E_BAD_ARG, "batch kernel must be 0 (f32 MFMA), 1 (bf16x3), 2 (f16) or 3 (auto)");
    for (auto &sh : idx->shards) {
      std::unique_lock<RwLock> xl(sh->mu);
      sh->batch_kernel = (int)value;
    }
    return TSH_OK;
  }
  return set_err(TSH_E_BAD_ARG, "unknown option %d", option);
}

int32_t tsh_bench_batch(tsh_index *idx, const float *queries, int32_t nq, int32_t k, int32_t iters,
                        double *out_avg_gemm_us, double *out_flops) {
  if (!idx || idx->shards.size() != 1 || !queries || nq <= 0 || k <= 0 || iters <= 0 || !out_avg_gemm_us)
    return set_err(TSH_E_BAD_ARG, "bad arguments");
  Shard *s = idx->shards[0].get();
  std::shared_lock<RwLock> sl = share(idx, s);
  if (s->rows == 0) return set_err(TSH_E_BAD_ARG, "empty index");
  if (s->safe_mode()) return set_err(TSH_E_BAD_ARG, "index is in safe mode: no batched path");
  int32_t entries = tsh_default_block_entries(k);
  std::vector<uint8_t> blocks((size_t)tsh_candidate_block_bytes(entries) * (size_t)nq);
  double acc = 0;
  for (int32_t i = 0; i < iters; ++i) {
    SearchOut so;
    so.h_blocks = blocks.data();
    // (the call tsh_search makes: results finalised on the device, lists as wide as that kernel takes -- with the
    // caller's 256-entry blocks an L2 corpus with varying norms overflowed here, and only here, and the overflows
    // talked the automatic kernel choice out of fp16 in the middle of a measurement)
    so.on_final = [](int32_t, int32_t, const char *, const int64_t *, const double *, const int32_t *) {};
    std::vector<int32_t> redo;
    s->batch->timed = true;  // (read under its mutex by the call below; measurement hooks are not run concurrently)
    int rc = shard_search_batch(s, s->batch, queries, nq, k, nullptr, entries, &so, &redo);
    s->batch->timed = false;
    if (rc) return rc;
    acc += s->batch->last_gemm_us;
  }
  *out_avg_gemm_us = acc / iters;
  if (out_flops) *out_flops = s->batch->last_flops;
  return TSH_OK;
}

// ---- probes of the pre-filter keys (tests/test_gpu_bands.py): what the kernels actually computed, next to the
// error bound the host claimed for it
int32_t tsh_probe_scan_keys(tsh_index *idx, const float *query, float *out_keys, float *out_eps_rel, float *out_delta_abs) {
  if (!idx || idx->shards.size() != 1 || !query || !out_keys || !out_eps_rel || !out_delta_abs)
    return set_err(TSH_E_BAD_ARG, "bad arguments");
  Shard *s = idx->shards[0].get();
  std::shared_lock<RwLock> sl = share(idx, s);
  if (s->rows == 0) return set_err(TSH_E_BAD_ARG, "empty index");
  Ctx *c = ctx_acquire(s, true);
  struct Rel {
    Shard *s;
    Ctx *c;
    ~Rel() { ctx_release(s, c); }
  } rel{s, c};
  int rc = ctx_prepare(s, c, tsh_default_block_entries(100), false);
  if (rc) return rc;
  hipStream_t st = s->aux_stream;
  memcpy(c->h_query, query, (size_t)s->dim * sizeof(float));
  for (int64_t j = s->dim; j < s->ld; ++j) c->h_query[j] = 0.f;
  const Band band = compute_band(s, c->h_query);
  if (band.force_all) return set_err(TSH_E_BAD_ARG, "the query / index is outside the error model (no band)");
  HIPCHK(hipMemcpyAsync(c->d_query, c->h_query, (size_t)s->ld * sizeof(float), hipMemcpyHostToDevice, st));
  static thread_local ScanArgsQ sa;
  const bool masked = !s->all_live;
  fill_scan_args(s, c, masked, false, &sa);
  launch_scan(sa, s->nch, s->metric, masked, st);
  std::vector<uint32_t> keys((size_t)s->rows);
  HIPCHK(hipMemcpyAsync(keys.data(), c->d_keys, keys.size() * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  for (int64_t i = 0; i < s->rows; ++i) out_keys[i] = keys[(size_t)i] >= KEY_NAN ? std::nanf("") : h_key2f(keys[(size_t)i]);
  *out_eps_rel = band.eps_rel;
  *out_delta_abs = band.delta_abs;
  return TSH_OK;
}

int32_t tsh_probe_batch_keys(tsh_index *idx, const float *queries, int32_t nq, int32_t k, float *out_keys, float *out_delta2) {
  if (!idx || idx->shards.size() != 1 || !queries || nq <= 0 || k <= 0 || !out_keys || !out_delta2)
    return set_err(TSH_E_BAD_ARG, "bad arguments");
  Shard *s = idx->shards[0].get();
  std::shared_lock<RwLock> sl = share(idx, s);
  if (s->rows == 0 || s->safe_mode()) return set_err(TSH_E_BAD_ARG, "empty index, or safe mode: no batched path");
  if ((int64_t)nq * s->rows > (1ll << 28)) return set_err(TSH_E_BAD_ARG, "nq x rows too large for a dense key matrix");
  const int32_t entries = tsh_default_block_entries(k);
  std::vector<uint8_t> blocks((size_t)tsh_candidate_block_bytes(entries) * (size_t)nq);
  SearchOut so;
  so.h_blocks = blocks.data();
  std::vector<int32_t> redo;
  BatchCtx *b = s->batch;
  b->last_sample_force = 1;  // (read under b->mu by the call below; probes are not run concurrently)
  int rc = shard_search_batch(s, b, queries, nq, k, nullptr, entries, &so, &redo);
  b->last_sample_force = 0;
  if (rc) return rc;
  std::lock_guard<std::mutex> lk(b->mu);
  if (b->last_sample != s->rows || b->last_nq != nq) return set_err(TSH_E_BUSY, "another batched call ran in between");
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipMemcpy2D(out_keys, (size_t)s->rows * 4, b->d_dense, (size_t)b->last_sample * 4, (size_t)s->rows * 4, (size_t)nq,
                     hipMemcpyDeviceToHost));
  if (s->split_grouped && s->batch_kernel_last.load() == 2) {  // a norm-grouped plane: the keys came out by plane position
    std::vector<uint32_t> perm((size_t)s->rows);
    HIPCHK(hipMemcpy(perm.data(), s->d_perm, (size_t)s->rows * 4, hipMemcpyDeviceToHost));
    std::vector<float> byrow((size_t)s->rows);
    for (int32_t q = 0; q < nq; ++q) {
      float *kq = out_keys + (size_t)q * (size_t)s->rows;
      for (int64_t p2 = 0; p2 < s->rows; ++p2) byrow[perm[(size_t)p2]] = kq[p2];
      memcpy(kq, byrow.data(), (size_t)s->rows * 4);
    }
  }
  // one bound for every row of the query: the shared part + the per-row part at the longest row
  const float *d2 = b->h_qaux + b->last_nq_pad, *al = b->h_qaux + 5 * (size_t)b->last_nq_pad;
  for (int32_t q = 0; q < nq; ++q) {
    const double v = (double)d2[q] + 2.0 * (double)al[q] * (double)s->max_norm * (1.0 + 1e-6);
    out_delta2[q] = (float)v;
    if ((double)out_delta2[q] < v) out_delta2[q] = std::nextafter(out_delta2[q], INFINITY);
  }
  return TSH_OK;
}

int32_t tsh_probe_batch_row_band(tsh_index *idx, int32_t nq, float *out_alpha2, float *out_beta2) {
  if (!idx || idx->shards.size() != 1 || nq <= 0 || !out_alpha2 || !out_beta2) return set_err(TSH_E_BAD_ARG, "bad arguments");
  Shard *s = idx->shards[0].get();
  std::shared_lock<RwLock> sl = share(idx, s);
  BatchCtx *b = s->batch;
  std::lock_guard<std::mutex> lk(b->mu);
  if (b->last_nq != nq || !b->h_qaux) return set_err(TSH_E_BUSY, "no batched call of that many queries to report on");
  const float *d2 = b->h_qaux + b->last_nq_pad, *al = b->h_qaux + 5 * (size_t)b->last_nq_pad;
  for (int32_t q = 0; q < nq; ++q) {
    out_alpha2[q] = 2.f * al[q];
    out_beta2[q] = d2[q];
  }
  return TSH_OK;
}

}  // extern "C"

#include "tsh_host_coldstart.inl.h"  // raw-vector file loader, tsh_index_open_ngh
#include "tsh_host_pq.inl.h"         // tsh_pq_train, tsh_index_pq_encode
#include "tsh_host_comm.inl.h"       // tsh_comm_*, tsh_search_sharded (RCCL)
