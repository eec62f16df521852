// tsh_host_batch.inl.h -- host side of the batched (matrix-core) path: scratch, band computation, launch sequence
// Part of the single translation unit tsh_lib.hip (textually included there; not compiled alone).

// ---- batched (matrix-core) path ---------------------------------------------------
#ifdef TSH_PROBES
static uint64_t *g_f16_dbg_buf = nullptr;  // TSH_F16_DBG & 32 probe (probe builds only: -DTSH_PROBES)
#endif
struct BatchCtx {
  std::mutex mu;  // one call at a time per scratch set (a shard has two: Shard::batch, batch2)
  float *d_Q = nullptr, *h_Q = nullptr;
  u32x4 *d_Qs = nullptr;  // bf16 planes of the padded queries
  int64_t qs_cap = 0;     // in u32x4 units
  float *d_qaux = nullptr, *h_qaux = nullptr;  // [qsq | delta2 | thr] x nq_pad
  int64_t q_cap = 0, aux_cap = 0, cc_cap = 0;  // element capacities
  float *d_dense = nullptr;
  int64_t dense_cap = 0;  // floats
  float *d_wnorm = nullptr;  // upper bounds of the sample rows' norms (per-row bands of the fp16 keys)
  int64_t wnorm_cap = 0;
  uint32_t *d_ck = nullptr, *d_cr = nullptr, *d_cc = nullptr;
  int64_t ck_cap = 0, cr_cap = 0;  // elements (nq * cand_cap wanted)
  uint8_t *d_blocks = nullptr, *h_blocks = nullptr;
  uint8_t *h_blocks_dev = nullptr;  // h_blocks as the device sees it (results are stored straight into it)
  uint32_t *d_final = nullptr;
  int64_t blocks_cap = 0, final_cap = 0;
  uint64_t *d_mask = nullptr, *h_mask = nullptr;
  int64_t mask_words = 0;
  float *d_dense2 = nullptr;  // the hub rows' dense keys (nq_pad x HUB_ROWS)
  int64_t dense2_cap = 0;
  // a call behind a SELECTIVE mask (listed mode, shard_search_batch): the kept rows' ids, their fp16 copy and norms
  uint32_t *d_list = nullptr, *h_list = nullptr;
  int64_t list_cap = 0;
  u32x4 *d_gplane = nullptr;
  int64_t gplane_cap = 0;  // in u32x4 units
  float *d_gsq = nullptr;
  int64_t gsq_cap = 0;
  hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr, e_done = nullptr;
  hipEvent_t e_chunk[3] = {nullptr, nullptr, nullptr};  // tail chunks but the last
  hipEvent_t e_up = nullptr;  // the call's inputs have arrived (upload stream -> batch stream)
  BlockEntry *d_quar_out = nullptr, *h_quar_out = nullptr;  // [query][quarantined row] exact sums
  int64_t quar_cap = 0;
  int64_t bytes = 0;
  double last_gemm_us = 0, last_flops = 0;  // key passes of the last TIMED call
  // timed: events around the key passes (tsh_bench_batch, TSH_TRACE_BATCH).  Off otherwise: every record between two
  // kernels is a packet the queue works through while the GPU idles (~6 us each, three of them per call)
  bool timed = false;
  int32_t est_backoff = 0;  // calls left during which the estimated threshold stays off (it failed too often)
  int64_t est_unverified = 0;  // queries whose estimated threshold turned out too tight (redone exactly)
  int64_t last_nq = 0, last_nq_pad = 0, last_sample = 0;  // shape of d_dense / h_qaux after the last call (tsh_probe_batch_keys)
  int32_t last_sample_force = 0;  // tsh_probe_batch_keys: the next call's dense sample covers every row
  double last_wait_us = 0;  // how long the previous call waited for the GPU after enqueueing
  std::vector<double> mag_a;  // per query of the current call: sum of q[i]^2 in element order
  // results finalised on the device (rerank_final_kernel): pinned host arrays the kernel stores into
  int64_t *h_fin_ids = nullptr, *fin_ids_dev = nullptr;   // nq x k
  double *h_fin_dist = nullptr, *fin_dist_dev = nullptr;  // nq x k
  int32_t *h_fin_cnt = nullptr, *fin_cnt_dev = nullptr;   // nq
  uint32_t *h_fin_info = nullptr, *fin_info_dev = nullptr;  // nq: flags << 24 | count of each list's header
  int64_t fin_info_cap = 0;
  int64_t fin_ids_cap = 0, fin_dist_cap = 0, fin_q_cap = 0;
  double *d_sqrt_mag = nullptr, *h_sqrt_mag = nullptr;    // per query: sqrt(mag_a)
  int64_t sqrt_mag_cap = 0;
};

void batch_free(BatchCtx *b) {
  hipFree(b->d_Q);
  hipFree(b->d_Qs);
  hipHostFree(b->h_Q);
  hipFree(b->d_qaux);
  hipHostFree(b->h_qaux);
  hipFree(b->d_dense);
  hipFree(b->d_wnorm);
  hipFree(b->d_ck);
  hipFree(b->d_cr);
  hipFree(b->d_cc);
  hipFree(b->d_blocks);
  hipHostFree(b->h_blocks);
  hipFree(b->d_final);
  hipFree(b->d_mask);
  hipHostFree(b->h_mask);
  hipFree(b->d_dense2);
  hipFree(b->d_list);
  hipHostFree(b->h_list);
  hipFree(b->d_gplane);
  hipFree(b->d_gsq);
  hipFree(b->d_quar_out);
  hipHostFree(b->h_quar_out);
  hipHostFree(b->h_fin_ids);
  hipHostFree(b->h_fin_dist);
  hipHostFree(b->h_fin_cnt);
  hipHostFree(b->h_fin_info);
  hipFree(b->d_sqrt_mag);
  hipHostFree(b->h_sqrt_mag);
  for (hipEvent_t e : {b->e0, b->e1, b->e2, b->e3, b->e_done, b->e_chunk[0], b->e_chunk[1], b->e_chunk[2], b->e_up})
    if (e) hipEventDestroy(e);
}

// TSH_TEST_FAIL_ALLOC_OVER=bytes (tests; obeyed only after TSH_OPT_TEST_HOOKS, see test_env): device allocations of the batched path at or above this size fail as if the
// device were full -- the degrade ladder of shard_search_any (planes -> f32 MFMA -> single-query scans) can then be
// walked deterministically, beside the test that really fills the device
inline bool alloc_fault(int64_t bytes) {
  const char *e = test_env("TSH_TEST_FAIL_ALLOC_OVER");  // (not cached: the hooks may be switched on after the first call)
  const int64_t over = e ? atoll(e) : 0;
  return over > 0 && bytes >= over;
}

// Is there room on the current device for an allocation of `bytes`?  Asked BEFORE hipMalloc for the batched path's big
// buffers: the runtime's own out-of-memory path is best not entered -- it walks its queues to release memory, and in
// a process that had created and destroyed CU-masked streams before (the test-suite; a host that opens and closes
// indexes) a failing 200 MB hipMalloc ended in a segmentation fault inside libhsa-runtime64 instead of
// hipErrorOutOfMemory (native backtrace: tools/abort_trace.sh).  Not a reservation -- another thread may take the room
// in between, and then hipMalloc's own verdict stands.
inline bool device_has_room(int64_t bytes) {
  if (bytes < (4ll << 20)) return true;  // (small buffers: not worth the query)
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
    (void)hipGetLastError();
    return true;
  }
  return (int64_t)free_b >= bytes + (16ll << 20);
}

template <typename T>
int regrow(T **dev, T **host, int64_t *cap, int64_t want, int64_t *bytes) {
  if (want <= *cap) return TSH_OK;
  hipFree(*dev);
  *dev = nullptr;
  if (host) {
    hipHostFree(*host);
    *host = nullptr;
  }
  *bytes -= *cap * (int64_t)sizeof(T);  // (a failed allocation below leaves an empty buffer, not a dangling capacity)
  *cap = 0;
  if (alloc_fault(want * (int64_t)sizeof(T))) return set_err(TSH_E_OOM, "hipMalloc failed: out of memory (injected)");
  if (!device_has_room(want * (int64_t)sizeof(T)))
    return set_err(TSH_E_OOM, "no room on the device for %lld bytes of batch scratch", (long long)(want * (int64_t)sizeof(T)));
  hipError_t e = hipMalloc(dev, (size_t)want * sizeof(T));
  if (e == hipSuccess && host) {
    e = hipHostMalloc(host, (size_t)want * sizeof(T), hipHostMallocDefault);
    if (e != hipSuccess) {
      hipFree(*dev);
      *dev = nullptr;
    }
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();  // the runtime's sticky copy: a later hipGetLastError() must not report this again
    return set_err(e == hipErrorOutOfMemory ? TSH_E_OOM : TSH_E_HIP, "hipMalloc of %lld bytes failed: %s",
                   (long long)(want * (int64_t)sizeof(T)), hipGetErrorString(e));
  }
  *bytes += want * (int64_t)sizeof(T);
  *cap = want;
  return TSH_OK;
}

// pinned host memory the device stores into (zero-copy): host pointer + the address the device uses
template <typename T>
int regrow_pinned(T **host, T **dev_view, int64_t *cap, int64_t want, int64_t *bytes) {
  if (want <= *cap) return TSH_OK;
  hipHostFree(*host);
  *host = nullptr;
  *dev_view = nullptr;
  *bytes -= *cap * (int64_t)sizeof(T);
  *cap = 0;
  HIPCHK(hipHostMalloc(host, (size_t)want * sizeof(T), hipHostMallocDefault));
  HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void **>(dev_view), *host, 0));
  *bytes += want * (int64_t)sizeof(T);
  *cap = want;
  return TSH_OK;
}

// (launch_batch_score_m: tsh_batch_tu.hip -- the key kernels compile as a translation unit of their own)

// Which order statistic of the sample the filter threshold is taken from (SampleSelArgs::k_est, tsh_batch.hip.h).
// The k smallest keys of all `rows` rows fall into the n_sample sample rows Binomial(k, n_sample / rows) at a time
// when rows are exchangeable; the p-th smallest sample key fails as a bound on the k-th smallest key overall only if
// p or more of them did.  p = the smallest value that makes this rarer than 1e-6 per query; B2 detects the failure
// and the query is redone, so a corpus that is NOT exchangeable (rows clustered by insertion time) costs time, never
// correctness -- and a call that sees more than a few failures switches the estimate off for the next calls.
int32_t batch_k_est(BatchCtx *b, int32_t k, int64_t rows, int64_t n_sample) {
  static const bool off = probe_env("TSH_BATCH_PROVEN_TAU") != nullptr && probe_env("TSH_BATCH_PROVEN_TAU")[0] == '1';
  if (off || n_sample >= rows || n_sample <= 0 || k <= 8 || b->est_backoff > 0) return k;
  const double pr = (double)n_sample / (double)rows;
  if (pr > 0.5) return k;  // (most of the rows are in the sample: its k-th key is as good as proven)
  // tail P(Binomial(k, pr) >= p) from p = k downwards
  std::vector<double> pmf((size_t)k + 1);
  const double lq = std::log1p(-pr), lp = std::log(pr);
  for (int i = 0; i <= k; ++i)
    pmf[(size_t)i] = std::exp(std::lgamma(k + 1.0) - std::lgamma(i + 1.0) - std::lgamma(k - i + 1.0) + i * lp + (k - i) * lq);
  double tail = 0;
  int p = k;
  for (; p >= 1; --p) {
    tail += pmf[(size_t)p];
    if (tail > 1e-6) break;
  }
  return std::min(k, std::max(p + 1, 4));
}

int64_t batch_sample_rows(int64_t rows, int32_t k, bool whole_blocks = false) {
  if (whole_blocks) {  // a norm-grouped plane: whole blocks of PG_ROWS positions (an unbiased sample of the norms)
    const int64_t n = round_up(batch_sample_rows(rows, k), PG_ROWS);
    return n >= rows ? rows : n;
  }
  if (rows <= 16384) return rows;
  static const int64_t div = probe_env("TSH_SAMPLE_DIV") ? std::max(4, atoi(probe_env("TSH_SAMPLE_DIV"))) : 32;  // experiments
  int64_t n = std::max<int64_t>(rows / div, (int64_t)k * rows / (3000 * div / 32));
  n = std::max<int64_t>(round_up(n, 256), 8192);  // whole row tiles of either tile shape
  n = std::max<int64_t>(n, round_up((int64_t)k * 4, 256));
  return std::min(n, rows);
}

// Sum of squares (f64, element order, one rounding per addition: == query_mag_a) and largest magnitude of up to
// QN_GROUP queries at once: a query's sum is one serial chain of additions, so a lone query waits out the add latency
// dim times; eight chains side by side keep the adder busy (1024 x 768: 85-100 -> 60-85 us of a call's preparation).
// ok[j] = every element of query j is within the error model (finite, |x| <= BIG_ABS).
constexpr int QN_GROUP = 8;
void query_norms(const float *q0, int64_t stride, int dim, int n, double *qn2, float *qmax, char *ok) {
  double acc[QN_GROUP] = {0, 0, 0, 0, 0, 0, 0, 0};
  float mx[QN_GROUP] = {0, 0, 0, 0, 0, 0, 0, 0};
  const float *p[QN_GROUP];
  for (int j = 0; j < QN_GROUP; ++j) p[j] = q0 + (int64_t)(j < n ? j : 0) * stride;
  for (int i = 0; i < dim; ++i)
    for (int j = 0; j < QN_GROUP; ++j) {
      const float v = p[j][i], a = std::fabs(v);
      mx[j] = a > mx[j] ? a : mx[j];  // (a NaN never becomes the maximum: caught below through the sum)
      acc[j] = acc[j] + (double)v * (double)v;
    }
  for (int j = 0; j < n; ++j) {
    qn2[j] = acc[j];
    qmax[j] = mx[j];
    ok[j] = mx[j] <= BIG_ABS && acc[j] == acc[j] && acc[j] < INFINITY;  // NaN / inf elements poison the sum
  }
}

inline double hld_f16(const Shard *s) { return (double)round_up(s->dim, 64); }  // padded reduction length of the fp16 planes

// The error band of a batched key, per query.  qn2: query_norms.
//   out_delta2   2 * (the part of the bound every key of the query shares), absolute, rounded up
//   out_alpha    fp16 keys of an L2 / inner-product index (round 5): the bound of the key of row v is
//                alpha |v| + delta2 / 2 + chain |thr'| -- the operand roundings act on the products, 2^-10 |q| |v| a row, so
//                a short row's key is known that much better than a long one's; 0 where every key has the same band
//   (chain: batch_chain(), per call; thr': the threshold the filtered pass starts its accumulators from)
bool batch_delta2(const Shard *s, int kernel, double qn2, float *out_delta2, float *out_qsq, float *out_alpha = nullptr) {
  const double qn = std::sqrt(qn2) * (1.0 + 1e-6), vmax = (double)s->max_norm * (1.0 + 1e-6);
  const double u2 = 1.1920928955078125e-07;        // 2^-23
  double gam = ((double)s->ld + 8.0) * u2;   // k-ordered fma chain of ld terms
  if (kernel == 1) {
    // three partial products per k accumulate in f32 (chain of 3 ld terms, each product exact),
    // and hi + lo drops 3.1 * 2^-18 |q_i||v_i| per element (tsh_batch.hip.h, bf16x3 variant)
    const double hld = (double)round_up(s->dim, 32);
    gam = (3.0 * hld + 8.0) * u2 * (1.0 + 0.00391) + 3.1 * 3.814697265625e-06;
  } else if (kernel == 2) {
    // both operands rounded to fp16 (2^-11 each, + their product), exact products accumulated in f32;
    // 2^-21: the f32 scaling / normalisation before the rounding; 2^-30: fp16 subnormal steps, which
    // sit >= 27 binades under the largest operand value after the power-of-two scaling
    const double hld = (double)round_up(s->dim, 64);
    // (a query is only batched when its largest element is within 2^8 of the batch's, see top_q)
    // The filtered cosine pass starts its accumulators at -theta (|theta| <= |q| (1 + 1e-2), capped by
    // BatchArgs::kmax) instead of zero: the chain carries twice the magnitude, and the key is formed from the
    // accumulator by one more subtraction and one multiplication by a power of two (tsh_batch_f16pp.hip.h)
    gam = 2.0 * (hld + 10.0) * u2 * (1.0 + 0.01) + 9.765625e-04 * (1.0 + 0.001) + 4.76837158203125e-07 +
          std::sqrt(hld) * 9.3e-10;
  }
  double delta, alpha = 0.0;
  if (kernel == 2 && s->metric != TSH_METRIC_COSINE) {
    // fp16 keys, L2 / inner product: per-row bands (round 5).  In key units the accumulator of (q, v) runs through
    //   L2:  thr' - |q|^2 - |v|^2 + alpha |v|  (its start)  ... + 2 q.v      inner product:  -thr' + alpha |v| ... + q.v
    // so its hld + 12 roundings (the chain's, the start value's fma, the key's subtraction) cost at most
    //   c P,  c = (hld + 12) u,  P = |thr'| + |q|^2 + |v|^2 + alpha |v| + 2 |q| |v| (1 + 2^-10)      (L2)
    //                            P = |thr'| + alpha |v| + |q| |v| (1 + 2^-10)                        (inner product)
    // and the operand roundings act on the products only: 2 e_op |q| |v| (L2), e_op |q| |v| (inner product).  Sorted
    // by what they multiply:  alpha |v|  (alpha appears on both sides: solved for)  +  c |thr'|  (known on the device
    // only: batch_chain)  +  the rest, the same for every row of the query.  Roundings of |q|^2, |v|^2 (f32 from f64
    // sums), of the seed and of the select kernels' key +- width arithmetic: 14 u A + 8 u kmax.
    const double hld = hld_f16(s), c = (hld + 12.0) * u2;
    const double e_op = 9.765625e-04 * (1.0 + 0.001) + 4.76837158203125e-07 + std::sqrt(hld) * 9.3e-10;
    const double pq = qn * (1.0 + 9.765625e-04);
    // (the fp16 subnormal steps are ABSOLUTE -- 2^-30 of the largest operand value, whatever the row's own norm: a short
    // row's small elements sit in them -- so that part of e_op goes with max|v| into the shared term)
    const double sub = std::sqrt(hld) * 9.3e-10 * qn * vmax;
    if (s->metric == TSH_METRIC_L2) {
      const double A = qn * qn + vmax * vmax, kmax = 1.01 * (qn + vmax) * (qn + vmax);
      alpha = (2.0 * e_op * qn + 2.0 * c * pq) / (1.0 - c);
      delta = c * A + 14.0 * u2 * A + 8.0 * u2 * kmax + 2.0 * sub;
    } else {
      const double kmax = 1.01 * qn * vmax;
      alpha = (e_op * qn + c * pq) / (1.0 - c);
      delta = 6.0 * u2 * qn * vmax + 8.0 * u2 * kmax + sub;
    }
    alpha *= 1.0 + 1e-6;
  } else if (s->metric == TSH_METRIC_IP) delta = gam * qn * vmax;
  else if (s->metric == TSH_METRIC_COSINE) delta = qn * (gam + 4.76837158203125e-07) * (kernel == 2 ? 1.0 + 1e-6 : 1.0);
  else delta = 2.0 * gam * qn * vmax + 6.0 * u2 * (qn * qn + vmax * vmax);
  delta += (double)s->dim * 7.5e-37;
  double d2 = 2.0 * delta * 1.0001;
  if (!(d2 < 1e30) || !(alpha < 1e30)) return false;
  *out_delta2 = (float)d2;
  if ((double)*out_delta2 < d2) *out_delta2 = std::nextafter(*out_delta2, INFINITY);
  *out_qsq = (float)qn2;
  if (out_alpha) {
    *out_alpha = (float)alpha;
    if ((double)*out_alpha < alpha) *out_alpha = std::nextafter(*out_alpha, INFINITY);
  }
  return true;
}
// fp16 keys of an L2 / inner-product index: c of batch_delta2 as the select kernels take it -- 2 c / (1 - 2 c), rounded
// up (the filter threshold thr' must still cover the k-th key once the band has grown by c |thr'| on either side); 0
// for every other key kernel / metric
float batch_chain2(const Shard *s, int kernel) {
  if (kernel != 2 || s->metric == TSH_METRIC_COSINE) return 0.f;
  const double c = (hld_f16(s) + 12.0) * 1.1920928955078125e-07;
  const double v = 2.0 * c / (1.0 - 2.0 * c) * (1.0 + 1e-6);
  float f = (float)v;
  if ((double)f < v) f = std::nextafter(f, INFINITY);
  return f;
}

inline bool trace_batch() {
  static const bool on = getenv("TSH_TRACE_BATCH") != nullptr;
  return on;
}

// ---- the hub rows (Shard::d_hub) -------------------------------------------------------------------------------------
constexpr int HUB_ROWS = 4096;  // sixteen 256-row tiles of the key kernel: a dense pass of 1 / 250 of a 1 M-row shard
// (Re)builds the gathered fp16 copy of the HUB_ROWS rows with the smallest (L2) / largest (inner product) norm.  Caller
// holds batch_enq_mu and s->mu shared; everything is enqueued on `st` but the selection, which needs the norms on the
// host once (4 MB for 1 M rows + a partial sort: a few milliseconds, at the first batched call and after the shard
// grew by a quarter).
int hub_build(Shard *s, hipStream_t st, int32_t hchunks, int v_exp) {
  const int64_t rows = s->rows;
  std::vector<float> sq((size_t)rows);
  HIPCHK(hipMemcpyAsync(sq.data(), s->d_sqnorm, (size_t)rows * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  std::vector<uint32_t> ids((size_t)rows);
  for (int64_t i = 0; i < rows; ++i) ids[(size_t)i] = (uint32_t)i;
  const bool shortest = s->metric == TSH_METRIC_L2;
  // (quarantined and absent rows carry |v|^2 = 0: under L2 they would crowd the hub -- they are never live, the dense
  // pass drops them -- so rows of norm zero go last either way)
  auto better = [&](uint32_t x, uint32_t y) {
    const float a = sq[x], b2 = sq[y];
    if ((a > 0.f) != (b2 > 0.f)) return a > 0.f;
    return shortest ? (a < b2 || (a == b2 && x < y)) : (a > b2 || (a == b2 && x < y));
  };
  std::nth_element(ids.begin(), ids.begin() + HUB_ROWS, ids.end(), better);
  ids.resize(HUB_ROWS);
  std::sort(ids.begin(), ids.end());  // (row order: neighbouring hub rows share pages)
  std::vector<float> hsq(HUB_ROWS);
  for (int i = 0; i < HUB_ROWS; ++i) hsq[(size_t)i] = sq[ids[(size_t)i]];
  const int64_t bytes = (int64_t)HUB_ROWS * hchunks * 64;
  if (!s->d_hub || s->hub_chunks != hchunks) {
    hipFree(s->d_hub);
    s->d_hub = nullptr;
    s->bytes -= s->hub_bytes;
    s->hub_bytes = 0;
    if (alloc_fault(bytes) || !device_has_room(bytes)) return set_err(TSH_E_OOM, "no room for the hub rows' copy");
    HIPCHK(hipMalloc(&s->d_hub, (size_t)bytes));
    if (!s->d_hub_ids) HIPCHK(hipMalloc(&s->d_hub_ids, (size_t)HUB_ROWS * 4));
    if (!s->d_hub_sq) HIPCHK(hipMalloc(&s->d_hub_sq, (size_t)HUB_ROWS * 4));
    s->hub_bytes = bytes + HUB_ROWS * 8;
    s->bytes += s->hub_bytes;
  }
  HIPCHK(hipMemcpyAsync(s->d_hub_ids, ids.data(), (size_t)HUB_ROWS * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(s->d_hub_sq, hsq.data(), (size_t)HUB_ROWS * 4, hipMemcpyHostToDevice, st));
  Half32Args ha{};
  ha.rows = s->d_rows;
  ha.inv_norm = nullptr;
  ha.out = s->d_hub;
  ha.ld = s->ld;
  ha.first = 0;
  ha.n = HUB_ROWS;
  ha.dim = s->dim;
  ha.kchunks = hchunks;
  ha.scale = std::ldexp(1.0f, v_exp);
  ha.ids = s->d_hub_ids;
  const int64_t total = (int64_t)HUB_ROWS * hchunks * 4;
  half_rows32_kernel<<<(unsigned)std::min<int64_t>((total + 255) / 256, 65536), 256, 0, st>>>(ha);
  HIPCHK(hipStreamSynchronize(st));  // (ids / hsq are this function's own vectors)
  HIPCHK(hipGetLastError());
  s->hub_rows_built = rows;
  s->hub_n = HUB_ROWS;
  s->hub_exp = v_exp;
  s->hub_chunks = hchunks;
  return TSH_OK;
}

// All nq queries in one pass over the rows on the matrix cores.  Queries the
// error model cannot cover, or whose lists overflow (ties), are reported in
// *redo and answered by the single-query path.  Caller holds s->mu shared.
// force_kernel >= 0: that key kernel instead of the handle's choice (0 = f32 MFMA on the rows as stored, the one that
// needs no converted copy: shard_search_any falls to it when the copy cannot be allocated).  An allocation failure
// returns TSH_E_OOM with nothing of the call left in flight and nothing written: the caller may retry or answer
// another way.
int shard_search_batch(Shard *s, BatchCtx *b, const float *queries, int32_t nq, int32_t k,
                       const MaskSrc &mask, int32_t entries, SearchOut *out, std::vector<int32_t> *redo,
                       int force_kernel = -1) {
  // the shard's first scratch set stands for "either": a call that finds it taken by a concurrent call uses the second
  std::unique_lock<std::mutex> lk(b->mu, std::defer_lock);
  if (b == s->batch && s->batch2 && !lk.try_lock()) {
    std::unique_lock<std::mutex> lk2(s->batch2->mu, std::try_to_lock);
    if (lk2.owns_lock()) {
      b = s->batch2;
      lk = std::move(lk2);
    }
  }
  if (!lk.owns_lock()) lk.lock();
  const double t_in = now_us();
  HIPCHK(hipSetDevice(s->device));
  const int64_t rows = s->rows, ld = s->ld;
  // A call the device finalises (below: gpu_final) never shows its candidate blocks to anybody, so their size is this
  // function's own business: as wide as the finalising kernel takes.  The caller's default (k + 156) is the
  // single-query path's, whose f32 keys have a band 100 x narrower than fp16 ones: an L2 / IP corpus whose norms
  // vary (band ~ max|v|, interesting rows ~ min|v|) overflowed 256-entry lists for a third of its queries, and every
  // overflow is a query redone by a whole scan (1 M x 768 L2, 1024 queries: 97 ms per call instead of 2.4).
  if (out->on_final && !out->d_blocks && s->quar_ids.empty() && entries <= RF_MAX && k <= entries) entries = RF_MAX;
  const size_t bb = (size_t)tsh_candidate_block_bytes(entries);
  // Workgroup tile (queries x rows).  f32 MFMA: 128 x 128.  bf16x3: 256 x 256 for batches of more than 128 queries,
  // 128 x 128 otherwise.  f16: 256 x 256 resp. 128 x 256.  Which key kernel runs is decided below (it needs the
  // scale of the rows); the padding of the batch only depends on the query side of the tile.
  static const int forced_tile = probe_env("TSH_BATCH_TILE") ? atoi(probe_env("TSH_BATCH_TILE")) : 0;  // experiments
  const int want_kernel = force_kernel >= 0 ? force_kernel : s->batch_kernel;
  const int32_t tile = (forced_tile == 128 || forced_tile == 256) && want_kernel != 0
                           ? forced_tile
                           : ((want_kernel != 0 && nq > 128) ? 256 : 128);  // (the f32 kernel has the 128 tile only)
  const int32_t nq_pad = (int32_t)round_up(nq, tile);
  // Sample size: the filtered pass keeps about k * rows / n_sample rows per query and
  // every survivor costs an atomic append, so the sample grows with k (survivors <= ~3000).
  // (the fp16 plane of an L2 / inner-product shard may be norm-grouped, plane_group_kernel: windows of whole blocks then.
  // Decided here from what is known before the key kernel is -- a superset; whole blocks suit every kernel)
  const bool want_group = s->batch_group && s->metric != TSH_METRIC_COSINE && want_kernel >= 2;
  int rc;
  // ---- the mask, early: how many rows it keeps decides how the call is scored ---------------------------------------
  const int32_t n_tiles_all = (int32_t)((rows + 63) / 64);
  // (a mask handle's words are resident, host and device: no copy of them in this call's scratch)
  if (mask.bytes && (rc = regrow(&b->d_mask, &b->h_mask, &b->mask_words, (int64_t)n_tiles_all, &b->bytes))) return rc;
  int64_t kept = -1;  // rows the mask keeps (tombstones not subtracted); -1: no mask
  if (mask.part) {
    kept = mask.part->kept;
  } else if (mask.bytes) {
    slice_mask(s, mask.bytes, b->h_mask, n_tiles_all);
    if (rows & 63) b->h_mask[n_tiles_all - 1] &= (1ull << (rows & 63)) - 1ull;  // (bits past the last row keep nothing)
    kept = popcount_words(b->h_mask, (size_t)n_tiles_all);
  }
  // LISTED mode (round 6): a mask that keeps at most LISTED_MAX rows, a small part of the shard.  Scoring all rows and
  // dropping all but a percent in the epilogue is the wrong way round: the kept rows' fp16 copy is GATHERED (their ids are
  // the mask as a list -- a handle's resident one, or this call's), and the dense pass scores just them: the "sample" is
  // every row there is, its threshold proven (k_est = k), and no filtered pass follows.  1 M x 768, keep 1 %: a
  // 64-query call 0.55 -> 0.2 ms, a 1024-query call 1.8 -> 0.3 ms.  fp16 keys only (the kernel that maps plane positions
  // to row ids); everything behind the candidate lists sees row ids, as ever.
  constexpr int64_t LISTED_MAX = 16384, LISTED_PART = 24;
  bool listed = false;
  if (kept >= 1 && kept <= LISTED_MAX && kept * LISTED_PART <= rows && !b->last_sample_force && want_kernel >= 2 &&
      (!mask.part || (mask.part->d_list && mask.part->list_padded > 0))) {
    bool f16_can = want_kernel == 2 || (want_kernel == 3 && s->metric == TSH_METRIC_COSINE);
    if (!f16_can && want_kernel == 3 && s->f16_denied_calls.load() <= 0) {  // the automatic choice's rule (below), read-only
      const double c_acc0 = (double)((s->dim + 63) / 64 + 12) * 1.1920928955078125e-07;
      f16_can = s->metric != TSH_METRIC_L2 ||
                (s->min_norm > 0.f && c_acc0 * (1.0 + (double)s->max_norm * (double)s->max_norm) <= 0.02 * (double)s->min_norm);
    }
    const float top0 = s->metric == TSH_METRIC_COSINE ? 1.0f : s->max_abs;
    int e0 = 0;
    if (top0 > 0.f) std::frexp(top0, &e0);
    listed = f16_can && 14 - e0 <= 55 && 14 - e0 >= -55;
  }
  const int64_t scan_rows = listed ? kept : rows;  // rows the key passes score
  const int64_t n_sample = listed ? kept : (b->last_sample_force ? rows : batch_sample_rows(rows, k, want_group));
  const int64_t ratio = scan_rows / std::max<int64_t>(n_sample, 1) + 1;
  int32_t k_est = batch_k_est(b, k, scan_rows, n_sample);  // (with a mask: recomputed below from the KEPT rows)
  const int32_t cand_cap = (int32_t)std::min<int64_t>(65536, std::max<int64_t>(4096, round_up(4 * (int64_t)k * ratio, 64)));
  if (!b->e0) {
    const unsigned wait_flag = blocking_wait() ? hipEventBlockingSync : 0;
    for (hipEvent_t *e : {&b->e0, &b->e1, &b->e2}) HIPCHK(hipEventCreate(e));
    HIPCHK(hipEventCreateWithFlags(&b->e3, wait_flag));  // timed, and the call's longest wait (the end of the key passes)
    HIPCHK(hipEventCreateWithFlags(&b->e_done, hipEventDisableTiming | wait_flag));
    for (hipEvent_t &e : b->e_chunk) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming | wait_flag));
    HIPCHK(hipEventCreateWithFlags(&b->e_up, hipEventDisableTiming));
  }
  if ((rc = regrow(&b->d_Q, &b->h_Q, &b->q_cap, (int64_t)nq_pad * ld, &b->bytes))) return rc;
  if ((rc = regrow(&b->d_qaux, &b->h_qaux, &b->aux_cap, (int64_t)nq_pad * 6, &b->bytes))) return rc;  // + thr, tau_est, kmax, alpha
  if ((rc = regrow(&b->d_dense, (float **)nullptr, &b->dense_cap, (int64_t)nq_pad * n_sample, &b->bytes))) return rc;
  if ((rc = regrow(&b->d_wnorm, (float **)nullptr, &b->wnorm_cap, n_sample, &b->bytes))) return rc;
  {
    // (each buffer keeps its own capacity: one that failed to grow must read as empty on the next call)
    const int64_t want = (int64_t)nq * cand_cap;
    if ((rc = regrow(&b->d_ck, (uint32_t **)nullptr, &b->ck_cap, want, &b->bytes))) return rc;
    if ((rc = regrow(&b->d_cr, (uint32_t **)nullptr, &b->cr_cap, want, &b->bytes))) return rc;
    if ((rc = regrow(&b->d_cc, (uint32_t **)nullptr, &b->cc_cap, (int64_t)nq_pad * CC_STRIDE, &b->bytes))) return rc;
  }
  if ((rc = regrow(&b->d_blocks, &b->h_blocks, &b->blocks_cap, (int64_t)nq * (int64_t)bb, &b->bytes))) return rc;
  HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void **>(&b->h_blocks_dev), b->h_blocks, 0));
  if ((rc = regrow(&b->d_final, (uint32_t **)nullptr, &b->final_cap, (int64_t)nq * entries, &b->bytes))) return rc;
  // The hub rows' bound (Shard::d_hub): fp16 keys of an L2 / inner-product shard whose norms differ at all, no caller
  // mask (the bound counts rows the mask may drop), big enough that sixteen more tiles are noise.  TSH_OPT_BATCH_HUB.
  bool use_hub = s->batch_hub && !mask && !listed && !b->last_sample_force && s->metric != TSH_METRIC_COSINE && rows >= 16 * HUB_ROWS &&
                 s->max_norm > 1.02f * s->min_norm && k <= HUB_ROWS / 4;

  // ---- bf16x3 / f16 kernels: keep the converted planes of the rows current ----------------------
  // auto: cosine keys are scale-free (unit rows, unit query), so fp16's fixed relative precision gives a
  // band that is narrow against ANY data; IP / L2 bands scale with the largest row norm, where bf16x3's
  // 25x tighter error keeps the candidate lists short when norms vary widely
  // -- unless the rows are nearly equal in norm (the usual normalised embeddings), when f16 serves them too
  // (a shard whose fp16 lists kept overflowing is served bf16x3 keys for a while -- F16_DENIAL_CALLS batched calls --
  // and then tried again: the corpus, or the calls' masks, may have changed)
  bool f16_ok = true;
  if (want_kernel == 3 && s->f16_denied_calls.load() > 0) {
    f16_ok = false;
    if (s->f16_denied_calls.fetch_sub(1) == 1) s->f16_strikes.store(0);
  }
  // Round 6: with a band PER ROW (alpha |v| + beta, batch_delta2) a row's own norm no longer
  // widens its neighbours' bands, so the automatic choice asks what is left -- the SHARED term.  Inner product: beta is
  // a few u |q| max|v|, nothing against the spacing of the keys that matter (the longest rows'): fp16 whatever the norms.
  // L2: beta ~ c (|q|^2 + max|v|^2) has to stay small against the spacing of the SHORT rows' keys, ~0.1 |q| min|v| near a
  // query's k-th neighbour: fp16 while c (1 + max|v|^2) <= 0.02 min|v| (|q| ~ 1 assumed; norms U(0.1, 3.2), a factor 32
  // apart: 4.8e-5 against 2e-3 -- 482 k queries/s where bf16x3 gives 199 k, tools/r6_wide_norms.sh; norms 2^-6 .. 2^6:
  // 1.2e-2 against 3e-4 -- every short row would be a candidate: bf16x3).  A shard whose fp16 lists overflow all the
  // same is moved to bf16x3 by the denial counter above.
  const double c_acc = (double)((s->dim + 63) / 64 + 12) * 1.1920928955078125e-07;
  const bool l2_ok = s->min_norm > 0.f && c_acc * (1.0 + (double)s->max_norm * (double)s->max_norm) <= 0.02 * (double)s->min_norm;
  const bool f16_fits = f16_ok && (s->metric != TSH_METRIC_L2 || l2_ok);
  int kern = want_kernel == 3 ? ((s->metric == TSH_METRIC_COSINE || f16_fits) ? 2 : 1) : want_kernel;
  if (listed) kern = 2;  // (decided above by the same rule; the scale's range was checked there too)
  int v_exp = 0;  // f16: rows are scaled by 2^v_exp so the largest magnitude lands in [2^13, 2^14)
  if (kern == 2) {
    const float top = s->metric == TSH_METRIC_COSINE ? 1.0f : s->max_abs;  // cosine planes hold unit rows
    int e = 0;
    if (top > 0.f) std::frexp(top, &e);  // top = m * 2^e, m in [0.5, 1)
    v_exp = 14 - e;
    if (v_exp > 55 || v_exp < -55) kern = 1;  // scales near the edge of f32's exponent range: bf16x3 instead
  }
  s->batch_kernel_last = kern;
  const bool use_bf16 = kern == 1, use_f16 = kern == 2, use_planes = use_bf16 || use_f16;
  // K-chunks of 32 for both plane formats; the f16 kernel walks its ring of four stages in whole turns, so its
  // planes are zero-padded to a multiple of four chunks
  const int32_t hchunks = use_f16 ? (int32_t)round_up((s->dim + 31) / 32, 4) : (int32_t)((s->dim + 31) / 32);
  // rows per workgroup tile: the f16 kernel's tile is 256 x 256 / 128 x 256 (tsh_batch_f16.hip.h), the others' square
  const int32_t tile_n = use_f16 ? 256 : tile;
  if (use_planes &&
      (rc = regrow(&b->d_Qs, (u32x4 **)nullptr, &b->qs_cap, round_up(nq_pad, PLANE_GROUP) * hchunks * 8, &b->bytes)))
    return rc;  // (sized for the bf16 hi + lo planes; the fp16 ones are half of it)
  if (listed) {  // the kept rows' fp16 copy (whole 256-row groups), their norms, their ids (a pointer mask's list: this call's)
    const int64_t prow = round_up(kept, PLANE_GROUP);
    if ((rc = regrow(&b->d_gplane, (u32x4 **)nullptr, &b->gplane_cap, prow * hchunks * 4, &b->bytes))) return rc;
    if ((rc = regrow(&b->d_gsq, (float **)nullptr, &b->gsq_cap, prow, &b->bytes))) return rc;
    if (!mask.part && (rc = regrow(&b->d_list, &b->h_list, &b->list_cap, kept + 64, &b->bytes))) return rc;
  }

  // ---- host prep: padded queries, per-query bands ------------------------------------
  float *h_qsq = b->h_qaux, *h_d2 = b->h_qaux + nq_pad, *h_kmax = b->h_qaux + 4 * (size_t)nq_pad;
  float *h_alpha = b->h_qaux + 5 * (size_t)nq_pad;  // per-row part of the band (fp16 keys, L2 / inner product; else 0)
  // the largest key a row can have (IP: -dot <= |q| max|v|; cosine planes hold unit rows; L2: (|q| + max|v|)^2): caps
  // the thresholds of the fp16 kernel's filtered pass.  1 % above it: a row AT the largest key still passes an
  // "everything passes" threshold with its key's error (<= 0.13 % of that scale) on top
  const double kmax_v = s->metric == TSH_METRIC_COSINE ? 1.0 : (double)s->max_norm;
  std::vector<char> bad((size_t)nq, 0);
  std::vector<float> qmax((size_t)nq_pad, 0.f);
  b->mag_a.resize((size_t)nq);  // the finaliser's sum of q[i]^2 (cosine), a by-product of the band computation
  parallel_for((nq_pad + QN_GROUP - 1) / QN_GROUP, [&](int32_t g) {
    const int32_t g0 = g * QN_GROUP, g1 = std::min(nq_pad, g0 + QN_GROUP), gn = std::max(0, std::min(nq, g1) - g0);
    char ok[QN_GROUP];
    if (gn > 0) query_norms(queries + (size_t)g0 * s->dim, s->dim, s->dim, gn, &b->mag_a[(size_t)g0], &qmax[(size_t)g0], ok);
    for (int32_t q = g0; q < g1; ++q) {
      float *dst = b->h_Q + (size_t)q * ld;
      h_kmax[q] = 0.f;
      h_alpha[q] = 0.f;
      if (q < nq && ok[q - g0] && batch_delta2(s, kern, b->mag_a[(size_t)q], &h_d2[q], &h_qsq[q], &h_alpha[q])) {
        memcpy(dst, queries + (size_t)q * s->dim, (size_t)s->dim * sizeof(float));
        for (int64_t i = s->dim; i < ld; ++i) dst[i] = 0.f;
        const double qn_q = std::sqrt(b->mag_a[(size_t)q]);
        const double km = (s->metric == TSH_METRIC_L2 ? (qn_q + kmax_v) * (qn_q + kmax_v) : qn_q * kmax_v) * 1.01 + 1e-30;
        h_kmax[q] = km < 3e38 ? (float)km : 3e38f;
      } else {
        if (q < nq) {
          qmax[(size_t)q] = 0.f;
          bad[(size_t)q] = 1;  // outside the error model: a zero query here, redone alone
        }
        memset(dst, 0, (size_t)ld * sizeof(float));
        h_d2[q] = 0.f;
        h_qsq[q] = 0.f;
        h_alpha[q] = 0.f;
      }
    }
  });
  // f16: one power-of-two scale for the whole batch; queries much smaller than the largest one would sit
  // in fp16's subnormal range, so they are answered alone
  float top_q = 0.f;
  int q_exp = 0;
  if (use_f16) {
    for (int32_t q = 0; q < nq; ++q) top_q = std::max(top_q, qmax[(size_t)q]);
    for (int32_t q = 0; q < nq; ++q)
      if (!bad[(size_t)q] && !(qmax[(size_t)q] >= top_q * 0.00390625f)) {
        bad[(size_t)q] = 1;
        memset(b->h_Q + (size_t)q * ld, 0, (size_t)ld * sizeof(float));
        h_d2[q] = 0.f;
        h_qsq[q] = 0.f;
      }
    if (top_q > 0.f) {
      int eq = 0;
      std::frexp(top_q, &eq);
      q_exp = 14 - eq;
    }
    if (q_exp > 55 || q_exp < -55) {  // same guard on the query side: answer them one by one
      q_exp = 0;
      for (int32_t q = 0; q < nq; ++q)
        if (!bad[(size_t)q]) {
          bad[(size_t)q] = 1;
          memset(b->h_Q + (size_t)q * ld, 0, (size_t)ld * sizeof(float));
          h_d2[q] = 0.f;
          h_qsq[q] = 0.f;
        }
    }
  }
  // (a pointer mask was sliced at the top)
  if (listed && !mask.part) list_mask_bits(b->h_mask, n_tiles_all, kept, b->h_list);  // ascending ids of the kept rows
  const uint64_t *h_mw = mask.part ? mask.part->h_words.data() : (mask.bytes ? b->h_mask : nullptr);  // this shard's slice,
  const uint64_t *d_mw = mask.part ? mask.part->d_words : (mask.bytes ? b->d_mask : nullptr);          // host and device
  // Where the sample sits.  The first n_sample rows serve any mask that keeps rows everywhere; a WHERE clause that
  // keeps one id RANGE leaves them without a single kept row -- no threshold, every kept row a survivor, every list
  // overflowing, every query redone by a scan of its own (1 M x 768, a 10 % range, 64-query calls: 5.5 k queries/s where
  // Bernoulli masks reach 120 k).  With a mask the window of n_sample consecutive rows (whole 256-row tiles) that keeps
  // the most rows is taken instead -- candidates every half window -- and the first one on a tie.  The bound needs
  // nothing of the sample but k kept rows inside it; an estimate it makes unrepresentative is caught by B2 as ever.
  int64_t s0 = 0;
  if (mask && !listed && n_sample < rows) {
    // in 64-row mask words; 256-row aligned (whole blocks of a norm-grouped plane: PG_ROWS-aligned)
    const int64_t wa = want_group ? PG_ROWS / 64 : 4;
    const int64_t wt = n_sample / 64, step = std::max<int64_t>(wa, wt / 2 / wa * wa);
    std::vector<int32_t> pre_own;
    if (!mask.part) {
      pre_own.assign((size_t)n_tiles_all + 1, 0);
      for (int32_t t = 0; t < n_tiles_all; ++t) pre_own[(size_t)t + 1] = pre_own[(size_t)t] + __builtin_popcountll(h_mw[t]);
    }
    const std::vector<int32_t> &pre = mask.part ? mask.part->pre : pre_own;  // (a handle counted its words when it was made)
    int64_t best = -1;
    for (int64_t w0 = 0; w0 * 64 + n_sample <= rows; w0 += step) {
      const int64_t kept = pre[(size_t)std::min<int64_t>(w0 + wt, n_tiles_all)] - pre[(size_t)w0];
      if (kept > best) {
        best = kept;
        s0 = w0 * 64;
      }
    }
    // the estimate's order statistic counts KEPT rows: a range mask may have half of them inside the window, where
    // n_sample / rows would say a thirtieth -- and a threshold taken that much too low fails its check for every query
    k_est = batch_k_est(b, k, std::max<int64_t>(pre[(size_t)n_tiles_all], 1), std::max<int64_t>(best, 0));
  }
  // quarantined rows (not live on the device): their exact sums for every query, added to the blocks below
  std::vector<uint32_t> quar_sel;
  const int32_t n_quar = (int32_t)s->quar_ids.size();
  if (n_quar) {
    quarantine_select(s, h_mw, &quar_sel);
    if (!quar_sel.empty() && !out->d_blocks) {
      if (!out->extra) return set_err(TSH_E_BAD_ARG, "no room for the quarantined rows' entries");
      if ((rc = regrow(&b->d_quar_out, &b->h_quar_out, &b->quar_cap, (int64_t)nq * n_quar, &b->bytes))) return rc;
    }
  }

  // The finaliser runs on the device too (rerank_final_kernel) when the caller wants final results on the host and
  // nothing but the device's own candidate lists goes into them
  const bool gpu_final = out->on_final && !out->d_blocks && quar_sel.empty() && entries <= RF_MAX && k <= entries;
  if (gpu_final) {
    if ((rc = regrow_pinned(&b->h_fin_ids, &b->fin_ids_dev, &b->fin_ids_cap, (int64_t)nq * k, &b->bytes))) return rc;
    if ((rc = regrow_pinned(&b->h_fin_dist, &b->fin_dist_dev, &b->fin_dist_cap, (int64_t)nq * k, &b->bytes))) return rc;
    if ((rc = regrow_pinned(&b->h_fin_cnt, &b->fin_cnt_dev, &b->fin_q_cap, (int64_t)nq, &b->bytes))) return rc;
    if ((rc = regrow_pinned(&b->h_fin_info, &b->fin_info_dev, &b->fin_info_cap, (int64_t)nq, &b->bytes))) return rc;
    if ((rc = regrow(&b->d_sqrt_mag, &b->h_sqrt_mag, &b->sqrt_mag_cap, (int64_t)nq, &b->bytes))) return rc;
    for (int32_t q = 0; q < nq; ++q) b->h_sqrt_mag[q] = std::sqrt(b->mag_a[(size_t)q]);
  }
  const double t_prep = now_us();
  bool grouped = false;  // the plane this call scores is norm-grouped (decided with the plane's upkeep, below)
  int n_chunks = 1;
  // chunk c of the tail = queries [chunk_q(c), chunk_q(c + 1)): the last chunk is the small one -- its finalisation
  // is all that is left to do on the host once the GPU is done
  auto chunk_q = [&](int c) -> int32_t {
    static const int cut4[5] = {0, 31, 62, 87, 100}, cut2[3] = {0, 60, 100};
    if (c <= 0) return 0;
    if (c >= n_chunks) return nq;
    const int pct = n_chunks == 4 ? cut4[c] : (n_chunks == 2 ? cut2[c] : 100 * c / n_chunks);
    return (int32_t)((int64_t)nq * pct / 100);
  };
  // ---- enqueue on the shard's batch stream (unmasked: the GEMM scales with CU count) ------
  // the inputs travel on the device's upload stream: with two calls in flight they arrive while the call in front
  // still computes (on the batch stream the 3 MB of a 1024 x 768 batch were 50-60 us of idle matrix cores per call:
  // two callers 450-600 k queries/s against 608-638 k on one box, tools/ab_batch.sh; one caller: no difference)
  {
    hipStream_t up = s->upload_stream;
    HIPCHK(hipMemcpyAsync(b->d_Q, b->h_Q, (size_t)nq_pad * ld * sizeof(float), hipMemcpyHostToDevice, up));
    HIPCHK(hipMemcpyAsync(b->d_qaux, b->h_qaux, (size_t)nq_pad * 2 * sizeof(float), hipMemcpyHostToDevice, up));
    HIPCHK(hipMemcpyAsync(b->d_qaux + 4 * (size_t)nq_pad, h_kmax, (size_t)nq_pad * 2 * sizeof(float), hipMemcpyHostToDevice, up));  // kmax, alpha
    if (mask.bytes) HIPCHK(hipMemcpyAsync(b->d_mask, b->h_mask, (size_t)n_tiles_all * 8, hipMemcpyHostToDevice, up));
    if (listed && !mask.part) HIPCHK(hipMemcpyAsync(b->d_list, b->h_list, (size_t)kept * 4, hipMemcpyHostToDevice, up));
    if (gpu_final) HIPCHK(hipMemcpyAsync(b->d_sqrt_mag, b->h_sqrt_mag, (size_t)nq * sizeof(double), hipMemcpyHostToDevice, up));
    HIPCHK(hipEventRecord(b->e_up, up));
  }
  {
    std::lock_guard<std::mutex> enq(s->batch_enq_mu);  // one call's sequence at a time on the one in-order stream
    hipStream_t st = s->batch_stream;
    if (use_planes && !listed) {  // (listed: the call scores a gathered copy of its own -- the shard's copy is neither needed nor built)
      const int64_t row_bytes = (int64_t)hchunks * (use_f16 ? 64 : 128);  // fp16: 2 B per element, bf16 hi + lo: 4 B
      if (s->split_mode != kern || (use_f16 && s->split_exp != v_exp)) s->split_valid = 0;  // other format / scale
      if (s->split_cap < s->cap || s->split_mode != kern) {  // first use, other format, or the row store grew
        if (s->d_split) hipFree(s->d_split);
        s->d_split = nullptr;
        s->bytes -= s->split_bytes;
        s->split_bytes = 0;
        s->split_cap = 0;
        s->split_valid = 0;
        const int64_t prow = round_up(s->cap, PLANE_GROUP);  // whole 256-row groups (plane_piece)
        // (no kernel of this call has been launched yet: a device too full for the copy is the caller's cue to score
        // with the f32 kernel instead, shard_search_any.  The call's inputs ARE on their way -- H2D copies out of the
        // pinned buffers the retry will rewrite -- so they are waited for before this call gives up)
        const hipError_t pe = alloc_fault(prow * row_bytes) || !device_has_room(prow * row_bytes)
                                  ? hipErrorOutOfMemory
                                  : hipMalloc(&s->d_split, (size_t)prow * (size_t)row_bytes);
        if (pe != hipSuccess) {
          s->d_split = nullptr;
          (void)hipGetLastError();
          (void)hipEventSynchronize(b->e_up);
          return set_err(pe == hipErrorOutOfMemory ? TSH_E_OOM : TSH_E_HIP, "hipMalloc of the %s copy of the rows (%lld bytes) failed: %s",
                         use_f16 ? "fp16" : "bf16x3", (long long)(prow * row_bytes), hipGetErrorString(pe));
        }
        s->split_cap = s->cap;
        s->split_bytes = prow * row_bytes;
        s->bytes += s->split_bytes;
      }
      s->split_mode = kern;
      s->split_exp = v_exp;
      grouped = use_f16 && want_group;
      if (grouped && s->perm_cap < s->split_cap) {  // (a device too full for the two maps: the plane stays in row order)
        hipFree(s->d_perm);
        hipFree(s->d_psq);
        s->d_perm = nullptr;
        s->d_psq = nullptr;
        s->bytes -= s->perm_bytes;
        s->perm_bytes = 0;
        s->perm_cap = 0;
        const int64_t prow = round_up(s->split_cap, PLANE_GROUP);
        if (alloc_fault(prow * 8) || !device_has_room(prow * 8) || hipMalloc(&s->d_perm, (size_t)prow * 4) != hipSuccess ||
            hipMalloc(&s->d_psq, (size_t)prow * 4) != hipSuccess) {
          (void)hipGetLastError();
          hipFree(s->d_perm);
          s->d_perm = nullptr;
          grouped = false;
        } else {
          s->perm_cap = s->split_cap;
          s->perm_bytes = prow * 8;
          s->bytes += s->perm_bytes;
        }
      }
      if (s->split_grouped != grouped) {  // the other order: the whole plane again
        s->split_valid = 0;
        s->split_grouped = grouped;
      }
    }
    use_hub = use_hub && use_f16 && !grouped;  // (either cure for crowded rows, not both)
    if (use_hub && (s->hub_rows_built < 0 || s->hub_exp != v_exp || s->hub_chunks != hchunks ||
                    rows - s->hub_rows_built > s->hub_rows_built / 4)) {
      // (a device too full for the copy, or any failure on the way: the call goes without the hub bound)
      if (hub_build(s, s->aux_stream, hchunks, v_exp) != TSH_OK) use_hub = false;
    }
    if (use_hub && regrow(&b->d_dense2, (float **)nullptr, &b->dense2_cap, (int64_t)nq_pad * HUB_ROWS, &b->bytes) != TSH_OK) use_hub = false;
    HIPCHK(hipStreamWaitEvent(st, b->e_up, 0));
    if (!quar_sel.empty() && !out->d_blocks) {
      QuarArgs qa{};
      qa.rows = s->d_rows;
      qa.Q = b->d_Q;
      qa.list = s->d_quar;
      qa.out = b->d_quar_out;
      qa.ld = ld;
      qa.ldq = ld;
      qa.row_base = s->row_base;
      qa.dim = s->dim;
      qa.cap = n_quar;
      qa.metric = s->metric;
      quarantine_kernel<<<dim3((unsigned)((n_quar + 63) / 64), (unsigned)nq), 64, 0, st>>>(qa);
      HIPCHK(hipMemcpyAsync(b->h_quar_out, b->d_quar_out, (size_t)nq * n_quar * sizeof(BlockEntry),
                            hipMemcpyDeviceToHost, st));
    }
    float *d_qsq = b->d_qaux, *d_d2 = b->d_qaux + nq_pad, *d_thr = b->d_qaux + 2 * (size_t)nq_pad;
    uint32_t *d_tau_est = reinterpret_cast<uint32_t *>(b->d_qaux + 3 * (size_t)nq_pad);
    BatchArgs a{};
    if (use_bf16) {
      auto split = [&](const float *src, int64_t first, int64_t n, u32x4 *dst) {
        SplitArgs sa{};
        sa.rows = src;
        sa.out = dst;
        sa.ld = ld;
        sa.first = first;
        sa.n = n;
        sa.dim = s->dim;
        sa.hchunks = hchunks;
        const int64_t total = n * hchunks * 4;
        split_rows_kernel<<<(unsigned)std::min<int64_t>((total + 255) / 256, 65536), 256, 0, st>>>(sa);
      };
      if (s->split_valid < rows) {
        split(s->d_rows, s->split_valid, rows - s->split_valid, s->d_split);
        s->split_valid = rows;
      }
      split(b->d_Q, 0, nq_pad, b->d_Qs);
      a.Qs = b->d_Qs;
      a.Vs = s->d_split;
      a.hchunks = hchunks;
    } else if (use_f16) {
      auto half = [&](const float *src, const float *inv, int64_t first, int64_t n, u32x4 *dst, int e, const uint32_t *ids = nullptr) {
        Half32Args ha{};
        ha.ids = ids;
        ha.rows = src;
        ha.inv_norm = inv;
        ha.out = dst;
        ha.ld = ld;
        ha.first = first;
        ha.n = n;
        ha.dim = s->dim;
        ha.kchunks = hchunks;
        ha.scale = std::ldexp(1.0f, e);
        const int64_t total = n * hchunks * 4;
        half_rows32_kernel<<<(unsigned)std::min<int64_t>((total + 255) / 256, 65536), 256, 0, st>>>(ha);
      };
      const uint32_t *d_ids = listed ? (mask.part ? mask.part->d_list : b->d_list) : nullptr;
      if (listed) {  // the kept rows, gathered: norms by list position, fp16 rows by list position
        gather_f32_kernel<<<(unsigned)((kept + 255) / 256), 256, 0, st>>>(s->d_sqnorm, d_ids, b->d_gsq, (int32_t)kept);
        half(s->d_rows, s->metric == TSH_METRIC_COSINE ? s->d_inv_norm : nullptr, 0, kept, b->d_gplane, v_exp, d_ids);
      } else if (s->split_valid < rows) {
        int64_t from = s->split_valid;
        if (grouped) {  // whole blocks again from the one the first stale row sits in: positions [from, rows) and their rows
          from = from / PG_ROWS * PG_ROWS;
          PlaneGroupArgs ga{};
          ga.sqnorm = s->d_sqnorm;
          ga.perm = s->d_perm;
          ga.psq = s->d_psq;
          ga.first = from;
          ga.rows = rows;
          ga.longest_first = s->metric == TSH_METRIC_IP ? 1 : 0;
          plane_group_kernel<<<(unsigned)((rows - from + PG_ROWS - 1) / PG_ROWS), 1024, 0, st>>>(ga);
        }
        half(s->d_rows, s->metric == TSH_METRIC_COSINE ? s->d_inv_norm : nullptr, from, rows - from, s->d_split, v_exp,
             grouped ? s->d_perm + from : nullptr);
        s->split_valid = rows;
      }
      half(b->d_Q, nullptr, 0, nq_pad, b->d_Qs, q_exp);
      a.Qs = b->d_Qs;
      a.Vs = listed ? b->d_gplane : s->d_split;
      a.row_ids = d_ids;
      a.hchunks = hchunks;
      a.dot_scale = std::ldexp(1.0f, -(q_exp + v_exp));
#ifdef TSH_PROBES
      static const int f16_dbg = probe_env("TSH_F16_DBG") ? atoi(probe_env("TSH_F16_DBG")) : 0;  // probes: results are wrong
      a.dbg = f16_dbg;
      static uint64_t *d_dbg = nullptr;
      if ((f16_dbg & 32) && !d_dbg) HIPCHK(hipMalloc(&d_dbg, (4 * 96 * 12 + 16) * sizeof(uint64_t)));
      if (f16_dbg & 32) HIPCHK(hipMemsetAsync(d_dbg, 0, (4 * 96 * 12 + 16) * sizeof(uint64_t), st));
      a.dbg_buf = d_dbg;
      g_f16_dbg_buf = d_dbg;
#endif
    }
    a.Q = b->d_Q;
    a.V = s->d_rows;
    a.inv_norm = use_f16 ? nullptr : s->d_inv_norm;  // f16 planes of a cosine corpus hold unit rows
    if (listed) {
      a.sqnorm = b->d_gsq;
    } else {
      a.sqnorm = grouped ? s->d_psq : s->d_sqnorm;  // (by plane position, like everything the key kernel indexes by column)
      a.row_ids = grouped ? s->d_perm : nullptr;
    }
    a.qsq = d_qsq;
    a.thr = d_thr;
    a.kmax = b->d_qaux + 4 * (size_t)nq_pad;
    // per-row bands: fp16 keys of an L2 / inner-product index (batch_delta2)
    // (the error-model probes, last_sample_force, read the dense pass's keys as they are: no widths there -- what that
    // call answers is discarded)
    const bool roww = use_f16 && s->metric != TSH_METRIC_COSINE && !b->last_sample_force;
    const float chain2 = batch_chain2(s, kern);
    a.alpha = roww ? b->d_qaux + 5 * (size_t)nq_pad : nullptr;
    // (listed: the list IS the mask; its rows' live bits are always looked up -- a handle's list may name rows deleted since)
    a.live = listed ? s->d_live : (s->all_live ? nullptr : s->d_live);
    a.mask = listed ? nullptr : d_mw;
    a.dense = b->d_dense;
    a.cand_key = b->d_ck;
    a.cand_row = b->d_cr;
    a.cand_cnt = b->d_cc;
    a.ld = ld;
    a.dense_ld = n_sample;
    a.nq = nq;
    a.nq_pad = nq_pad;
    a.kchunks = (int32_t)((ld + BT_K - 1) / BT_K);
    a.cand_cap = cand_cap;
    a.tile_m = tile;
    a.q_tiles = nq_pad / tile;
    // B0: dense keys of the sample rows
    a.row0 = (int32_t)s0;
    a.row1 = (int32_t)(s0 + n_sample);
    a.n_tiles = (int32_t)((n_sample + tile_n - 1) / tile_n);
    const bool timed = b->timed || trace_batch();
    if (timed) HIPCHK(hipEventRecord(b->e0, st));
    launch_batch_score_m(s->metric, a, true, st, s->cus);
    if (use_hub) {  // the hub rows, densely: a gathered copy with its own norms and a map back to the rows' live bits
      BatchArgs ah = a;
      ah.Vs = s->d_hub;
      ah.sqnorm = s->d_hub_sq;
      ah.row_ids = s->d_hub_ids;
      ah.dense = b->d_dense2;
      ah.dense_ld = HUB_ROWS;
      ah.row0 = 0;
      ah.row1 = HUB_ROWS;
      ah.n_tiles = HUB_ROWS / tile_n;
      launch_batch_score_m(s->metric, ah, true, st, s->cus);
    }
    if (timed) HIPCHK(hipEventRecord(b->e1, st));
    // B0s: per-query threshold + the sample's own candidates
    SampleSelArgs ss{};
    ss.dense = b->d_dense;
    ss.delta2 = d_d2;
    ss.thr = d_thr;
    ss.cand_key = b->d_ck;
    ss.cand_row = b->d_cr;
    ss.cand_cnt = b->d_cc;
    ss.dense_ld = n_sample;
    ss.n_sample = (int32_t)n_sample;
    ss.k = k;
    ss.k_est = k_est;
    ss.tau_est = d_tau_est;
    ss.cand_cap = cand_cap;
    ss.row0 = (int32_t)s0;
    ss.alpha = a.alpha;
    ss.sqnorm = s->d_sqnorm;
    ss.chain2 = chain2;
    ss.norm_max = s->max_norm;
    ss.hub_dense = use_hub ? b->d_dense2 : nullptr;
    ss.hub_ld = HUB_ROWS;
    ss.hub_n = HUB_ROWS;
    ss.row_ids = a.row_ids;
    if (roww) {  // the sample rows' norms, once per call
      sample_norms_kernel<<<(unsigned)((n_sample + 255) / 256), 256, 0, st>>>(a.sqnorm + s0, b->d_wnorm, (int32_t)n_sample);
      ss.wnorm = b->d_wnorm;
    }
    // one workgroup per query: a small batch leaves most CUs empty, so its workgroups get sixteen waves instead of
    // four (128 queries: 76 -> 54 and 58 -> 50 us); with a thousand workgroups the wide shape loses (the one-wave
    // bisect phases idle fifteen waves: 103 -> 235 us)
    const bool wide_select = nq <= 256;
    if (roww) {
      if (wide_select) batch_sample_select_roww_kernel<BS_THREADS_WIDE><<<nq, BS_THREADS_WIDE, 0, st>>>(ss);
      else batch_sample_select_roww_kernel<BS_THREADS><<<nq, BS_THREADS, 0, st>>>(ss);
    } else if (wide_select) batch_sample_select_kernel<BS_THREADS_WIDE><<<nq, BS_THREADS_WIDE, 0, st>>>(ss);
    else batch_sample_select_kernel<BS_THREADS><<<nq, BS_THREADS, 0, st>>>(ss);
    // B1: everything else, filtered
    if (timed) HIPCHK(hipEventRecord(b->e2, st));
    if (s0 > 0 && !listed) {  // (a sample window inside the rows: the filtered pass runs on either side of it)
      a.row0 = 0;
      a.row1 = (int32_t)s0;
      a.n_tiles = (int32_t)((s0 + tile_n - 1) / tile_n);
      launch_batch_score_m(s->metric, a, false, st, s->cus);
    }
    if (!listed && rows > s0 + n_sample) {
      a.row0 = (int32_t)(s0 + n_sample);
      a.row1 = (int32_t)rows;
      a.n_tiles = (int32_t)((rows - s0 - n_sample + tile_n - 1) / tile_n);
      launch_batch_score_m(s->metric, a, false, st, s->cus);
    }
    HIPCHK(hipEventRecord(b->e3, st));
    // B2 + rerank
    FinalSelArgs fs{};
    fs.cand_key = b->d_ck;
    fs.cand_row = b->d_cr;
    fs.cand_cnt = b->d_cc;
    fs.delta2 = d_d2;
    fs.tau_est = d_tau_est;
    // host mode: headers and re-ranked entries are stored straight into pinned host memory (only the `count`
    // valid entries cross PCIe, and no copy sits between the last kernel and the host)
    const bool zero_copy = out->d_blocks == nullptr;
    fs.blocks = b->d_blocks;
    // (device-finalised calls: the re-rank's launch reports each header in one word beside the results; the headers'
    // own stores into host memory, 1024 small PCIe writes in front of that launch, are left out)
    fs.blocks_host = zero_copy && !gpu_final ? b->h_blocks_dev : nullptr;
    fs.final_rows = b->d_final;
    fs.block_bytes = (int64_t)bb;
    fs.row_base = s->row_base;
    fs.shard_rows = rows;
    fs.k = k;
    fs.cand_cap = cand_cap;
    fs.entries = entries;
    fs.metric = s->metric;
    fs.alpha = a.alpha;
    fs.sqnorm = s->d_sqnorm;
    fs.thr = d_thr;
    fs.kmax = a.kmax;
    fs.chain2 = chain2;
    if (roww) {
      if (wide_select) batch_final_select_roww_kernel<BS_THREADS_WIDE><<<nq, BS_THREADS_WIDE, 0, st>>>(fs);
      else batch_final_select_roww_kernel<BS_THREADS><<<nq, BS_THREADS, 0, st>>>(fs);
    } else if (wide_select) batch_final_select_kernel<BS_THREADS_WIDE><<<nq, BS_THREADS_WIDE, 0, st>>>(fs);
    else batch_final_select_kernel<BS_THREADS><<<nq, BS_THREADS, 0, st>>>(fs);
    RerankBatchArgs rb{};
    rb.rows = s->d_rows;
    rb.Q = b->d_Q;
    rb.final_rows = b->d_final;
    rb.blocks = b->d_blocks;
    rb.out_blocks = zero_copy ? b->h_blocks_dev : b->d_blocks;
    rb.block_bytes = (int64_t)bb;
    rb.ld = ld;
    rb.row_base = s->row_base;
    rb.dim = s->dim;
    rb.entries = entries;
    rb.metric = s->metric;
    // the tail runs in chunks of queries: while the GPU re-ranks chunk c + 1, the host already finalises chunk c
    // (out->on_chunk).  (One launch over all chunks, each reporting through a pinned word when its last workgroup
    // has stored its entries, was tried: 167 us against 4 x 48 -- no overlap to speak of -- and the waits for the
    // host-memory stores to land doubled it.  Consecutive chunks on two streams: no change, a chunk's 512 workgroups of
    // 70 KB LDS fill every CU and the next chunk's only start as they leave.)
    n_chunks = gpu_final || !out->on_chunk ? 1 : (nq >= 512 ? 4 : (nq >= 256 ? 2 : 1));  // (256 queries: 1.00 -> 0.94 ms; 128: no gain)
    if (gpu_final) {
      // one launch for the whole call: a re-ranking wave is latency-bound, so the more of them are in flight the better,
      // and with the finaliser on the device no host work is left to overlap chunk by chunk
      RerankFinalArgs rf{};
      rf.r = rb;
      rf.r.q0 = 0;
      rf.sqrt_mag_a = b->d_sqrt_mag;
      rf.thr = out->fin_thr;
      rf.out_ids = b->fin_ids_dev;
      rf.out_dist = b->fin_dist_dev;
      rf.out_count = b->fin_cnt_dev;
      rf.out_info = b->fin_info_dev;
      rf.k = k;
      static const bool old_shape = probe_env("TSH_RF_SHAPE") != nullptr && probe_env("TSH_RF_SHAPE")[0] == '2';  // A/B: two waves, 64-float pieces
      if (old_shape) rerank_final_kernel<RwBig><<<(unsigned)nq, 64 * RwBig::WAVES, 0, st>>>(rf);
      else rerank_final_kernel<RwFin><<<(unsigned)nq, 64 * RwFin::WAVES, 0, st>>>(rf);
      HIPCHK(hipEventRecord(b->e_done, st));
    } else
    for (int c = 0; c < n_chunks; ++c) {
      const int32_t q0 = chunk_q(c), q1 = chunk_q(c + 1);
      rb.q0 = q0;
      rerank_batch_kernel<<<dim3((unsigned)((entries + RW_CAND - 1) / RW_CAND), (unsigned)(q1 - q0)), 64 * RwBig::WAVES, 0, st>>>(rb);
      if (!quar_sel.empty() && out->d_blocks) {  // shard mode: the quarantined rows go into the device blocks
        QuarAppendArgs qa{};
        qa.rows = s->d_rows;
        qa.Q = b->d_Q + (size_t)q0 * ld;
        qa.list = s->d_quar;
        qa.mask = d_mw;
        qa.blocks = b->d_blocks + (size_t)q0 * bb;
        qa.ld = ld;
        qa.ldq = ld;
        qa.row_base = s->row_base;
        qa.block_bytes = (int64_t)bb;
        qa.dim = s->dim;
        qa.entries = entries;
        qa.metric = s->metric;
        quarantine_append_kernel<<<dim3((unsigned)((n_quar + 63) / 64), (unsigned)(q1 - q0)), 64, 0, st>>>(qa);
      }
      if (!zero_copy)
        HIPCHK(hipMemcpyAsync(b->h_blocks + (size_t)q0 * bb, b->d_blocks + (size_t)q0 * bb, (size_t)(q1 - q0) * bb,
                              hipMemcpyDeviceToHost, st));
      HIPCHK(hipEventRecord(c + 1 < n_chunks ? b->e_chunk[c] : b->e_done, st));
    }
  }
  const double t_enq = now_us();
  // the finalisation below starts the moment the GPU is done: if that is within a millisecond (going by the
  // previous call), the pool's workers keep polling through the wait instead of parking (waking them cost 50-70 us
  // of a 128-query call's 90 us of finalisation)
  if ((out->on_chunk || gpu_final) && b->last_wait_us > 0 && b->last_wait_us < 900.0)
    HostPool::get().stay_awake_until(t_enq + b->last_wait_us * 1.1 + 50.0);
  std::vector<char> skip((size_t)nq, 0);
  int32_t n_unverified = 0, n_band_over = 0;
  double t_gpu = 0;
  // From the end of the key passes on the pool gets a job every few dozen microseconds (one per chunk of the tail):
  // the long wait blocks, then the workers are woken and poll until the call is over
  // (ranks that outnumber the CPU quota: a device-finalised call has one copy left for the pool -- not worth workers
  // polling through the select and re-rank kernels)
  std::unique_ptr<HostPool::Hold> hold;
  if ((out->on_chunk || gpu_final) && nq >= 24 && !(gpu_final && blocking_wait())) {
    HIPCHK(hipEventSynchronize(b->e3));
    hold.reset(new HostPool::Hold());
  }
  for (int c = 0; c < n_chunks; ++c) {
    const int32_t q0 = chunk_q(c), q1 = chunk_q(c + 1);
    HIPCHK(hipEventSynchronize(c + 1 < n_chunks ? b->e_chunk[c] : b->e_done));
    if (c == 0) b->last_wait_us = now_us() - t_enq;
    if (c + 1 == n_chunks) {
      HIPCHK(hipGetLastError());
      t_gpu = now_us();
    }
    for (int32_t q = q0; q < q1; ++q) {
      const BlockHeader *h = reinterpret_cast<const BlockHeader *>(b->h_blocks + (size_t)q * bb);
      const uint32_t h_flags = gpu_final ? b->h_fin_info[q] >> 24 : h->flags;
      const uint32_t h_count = gpu_final ? b->h_fin_info[q] & 0xFFFFFFu : h->count;
      if (bad[(size_t)q] || (h_flags & FLAG_LIST_OVERFLOW)) {
        if (h_flags & FLAG_TAU_UNVERIFIED) ++n_unverified;
        else if (!bad[(size_t)q]) ++n_band_over;  // a threshold existed and held: the band itself was too wide for the list
        redo->push_back(q);
        skip[(size_t)q] = 1;
      } else {
        s->c_cands += h_count;
        if (!quar_sel.empty() && !out->d_blocks) {  // the quarantined rows join this query's candidates
          std::vector<BlockEntry> &ex = (*out->extra)[(size_t)(out->q_base + q)];
          ex.clear();
          for (uint32_t i : quar_sel) ex.push_back(b->h_quar_out[(size_t)q * n_quar + i]);
        }
      }
    }
    const double t_c0 = now_us();
    if (gpu_final) {
      out->on_final(q0, q1, skip.data(), b->h_fin_ids, b->h_fin_dist, b->h_fin_cnt);  // a copy is all that is left
    } else if (out->on_chunk) {
      out->on_chunk(q0, q1, skip.data(), b->h_blocks, b->mag_a.data());  // finalised straight from the pinned buffer
      // (queries the single-query path will redo write their blocks to out->h_blocks themselves)
    } else if (out->h_blocks) {
      memcpy(out->h_blocks + (size_t)q0 * bb, b->h_blocks + (size_t)q0 * bb, (size_t)(q1 - q0) * bb);
    }
    if (trace_batch()) fprintf(stderr, "[tsh batch]   chunk %d: event at %.0f us, finalised in %.0f us\n", c, t_c0 - t_enq, now_us() - t_c0);
  }
  if (b->timed || trace_batch()) {
    float ms0 = 0.f, ms1 = 0.f;
    HIPCHK(hipEventElapsedTime(&ms0, b->e0, b->e1));
    HIPCHK(hipEventElapsedTime(&ms1, b->e2, b->e3));
    b->last_gemm_us = ((double)ms0 + (double)ms1) * 1e3;
    b->last_flops = 2.0 * nq * (double)scan_rows * (double)s->dim;
  }
  // auto mode: fp16 keys whose band keeps overflowing the lists of this corpus (every overflow is a whole scan) give
  // way to bf16x3 ones, 25 x narrower, after two such calls (cosine keys are scale-free: never)
  // Counted: queries whose threshold held and whose list still overflowed.  Not counted: queries the error model
  // rejected (bad[]), failed estimates, and calls with a caller mask (a mask that leaves the sample fewer than k kept
  // rows has no threshold at all: every kept row survives, whatever the key kernel).
  if (kern == 2 && want_kernel == 3 && s->metric != TSH_METRIC_COSINE && entries >= RF_MAX && !mask && !b->last_sample_force) {  // (at the widest lists only)
    if (n_band_over > std::max(2, nq / 16)) {
      if (s->f16_strikes.fetch_add(1) + 1 >= 2) s->f16_denied_calls.store(F16_DENIAL_CALLS);
    } else if (n_band_over == 0) {
      s->f16_strikes.store(0);
    }
  }
  b->est_unverified += n_unverified;
  if (b->est_backoff > 0) --b->est_backoff;
  else if (n_unverified > std::max(1, nq / 64)) b->est_backoff = 64;  // the sample does not represent these rows
  b->last_nq = nq;
  b->last_nq_pad = nq_pad;
  b->last_sample = n_sample;
  s->c_batches++;
  s->c_searches += nq - (int64_t)redo->size();
  if (out->d_blocks) {
    hipStream_t us = out->user_stream ? out->user_stream : s->batch_stream;
    HIPCHK(hipMemcpyAsync(out->d_blocks, b->d_blocks, (size_t)nq * bb, hipMemcpyDeviceToDevice, us));
    if (blocking_wait()) {  // (hipStreamSynchronize spins)
      HIPCHK(hipEventRecord(b->e_done, us));
      HIPCHK(hipEventSynchronize(b->e_done));
    } else {
      HIPCHK(hipStreamSynchronize(us));
    }
  }
#ifdef TSH_PROBES
  if (g_f16_dbg_buf && nq >= 1024) {  // probe: step timeline of two waves of workgroup 0 (main pass: the last launch)
    std::vector<uint64_t> hb(4 * 96 * 12 + 16);
    HIPCHK(hipMemcpy(hb.data(), g_f16_dbg_buf, hb.size() * 8, hipMemcpyDeviceToHost));
    fprintf(stderr, "[pp cnt] wave tiles %llu, crowded %llu, more than two per lane %llu, hit blocks %llu\n", (unsigned long long)hb[4 * 96 * 12],
            (unsigned long long)hb[4 * 96 * 12 + 1], (unsigned long long)hb[4 * 96 * 12 + 2], (unsigned long long)hb[4 * 96 * 12 + 3]);
    static const bool raw = probe_env("TSH_F16_GEN") == nullptr || probe_env("TSH_F16_GEN")[0] != '2';
    if (raw) {  // ping-pong kernel: raw stamps relative to the wave's first, one line per chunk (points 0-9)
      const int wv[4] = {0, 4, 1, 5};
      for (int w = 0; w < 4; ++w) {
        const uint64_t *b0 = &hb[(size_t)(w * 96) * 12];
        fprintf(stderr, "[pp dbg] wave %d hw_id 0x%llx simd %llu\n", wv[w], (unsigned long long)b0[11], (unsigned long long)((b0[11] >> 4) & 3));
        for (int st2 = 0; st2 < 60; ++st2) {
          const uint64_t *t = &hb[(size_t)(w * 96 + st2) * 12];
          fprintf(stderr, "[pp dbg]  w%d c%02d:", wv[w], st2);
          for (int pt = 0; pt < 12; ++pt) fprintf(stderr, " %7lld", t[pt] ? (long long)(t[pt] - hb[0]) : -1ll);
          fprintf(stderr, "\n");
        }
      }
    } else
    for (int w = 0; w < 2; ++w) {
      fprintf(stderr, "[f16 dbg] wave %d: per step: wait vmcnt | barrier | DMA issue | reads + slab 0 + reads + slab 1 | - | (to next step)\n", w * 4);
      for (int st2 = 20; st2 < 52; ++st2) {
        const uint64_t *t = &hb[(size_t)(w * 96 + st2) * 12], *tn = &hb[(size_t)(w * 96 + st2 + 1) * 12];
        fprintf(stderr, "[f16 dbg]   step %2d: %5lld %5lld %5lld %5lld+%5lld %5lld | %5lld   (epilogue stamp %lld)\n", st2, (long long)(t[1] - t[0]),
                (long long)(t[2] - t[1]), (long long)(t[3] - t[2]), (long long)(t[7] - t[3]), (long long)(t[4] - t[7]), (long long)(t[5] - t[4]),
                (long long)(tn[0] - t[5]), (long long)(t[6] ? t[6] - t[5] : 0));
        if (t[6]) fprintf(stderr, "[f16 dbg]     epilogue before this step: settle %lld, walk %lld, list -> atomics %lld, to the step %lld\n",
                          (long long)(t[8] - t[6]), (long long)(t[9] - t[8]), (long long)(t[10] - t[9]), (long long)(t[0] - t[10]));
      }
    }
  }
#endif
  if (trace_batch())
    fprintf(stderr, "[tsh batch] nq=%d prep %.0f us, enqueue %.0f us, gpu wait %.0f us, post %.0f us (gemm %.0f us)\n", nq,
            t_prep - t_in, t_enq - t_prep, t_gpu - t_enq, now_us() - t_gpu, b->last_gemm_us);
  return TSH_OK;
}

// rows a caller's mask keeps on this shard (the pointer form, counted for the cost model below: 125 KB at 1 M rows, ~5 us)
int64_t mask_kept_rows(const Shard *s, const uint8_t *bytes) {
  const int64_t lo = s->row_base, hi = s->row_base + s->rows;
  if (hi <= lo) return 0;
  int64_t n = 0, i = lo;
  for (; i < hi && (i & 7); ++i) n += (bytes[i >> 3] >> (i & 7)) & 1;
  const int64_t whole = (hi - i) / 64;
  for (int64_t w = 0; w < whole; ++w) {
    uint64_t v;
    memcpy(&v, bytes + (i >> 3) + 8 * w, 8);
    n += __builtin_popcountll(v);
  }
  for (i += 64 * whole; i < hi; ++i) n += (bytes[i >> 3] >> (i & 7)) & 1;
  return n;
}

// does a call of nq queries on this shard go to the matrix cores?  kept: rows a mask of the call keeps (-1: none / unknown)
bool shard_takes_batch(const Shard *s, int32_t batch_min_nq, int32_t nq, int32_t k, int64_t kept = -1) {
  // batch_min_nq == 1: decide by cost.  Measured: a batched call costs about 0.30 ms plus one
  // pass over the converted rows at ~4 TB/s, whatever nq <= 128 is; pipelined single-query scans cost one
  // pass over the f32 rows at ~6.6 TB/s plus ~25 us each.  At 1 M x 768 that is 0.67 vs 1.04 ms for TWO queries.
  // With a mask (round 6) both sides change: a query of its own reads the kept rows only -- from their exact sums in two
  // dispatches when there are at most exact_rows of them (16.7 us per query at 10 k kept rows of 768 floats, pipelined),
  // by a masked scan otherwise (57-65 us at 100 k) --, and a batched call behind a selective mask scores a gathered
  // copy of the kept rows (listed mode, shard_search_batch: ~0.2 ms whatever nq <= 128 is).  Eight queries behind a 1 %
  // mask of 1 M rows: 0.13 ms one by one, 0.2 ms batched (0.55 before the listed mode) -- the model used to send
  // every masked call of two queries or more to the matrix cores.
  bool enough = batch_min_nq > 1 && nq >= batch_min_nq;
  if (batch_min_nq == 1 && nq >= 2) {
    const int kern = s->batch_kernel == 3 ? 2 : s->batch_kernel;
    const double plane_b = kern == 0 ? 4.0 : (kern == 1 ? 4.0 : 2.0);
    double t_batch = 300.0 + (double)s->rows * (double)s->dim * plane_b / 4.0e6 * ((nq + 127) / 128);
    double t_single = (double)nq * ((double)s->rows * (double)s->ld * 4.0 / 6.6e6 + 25.0);
    if (kept >= 0 && kept < s->rows) {
      const double kept_bytes = (double)std::max<int64_t>(kept, 1) * (double)s->ld * 4.0;
      if (kept <= std::min<int64_t>(s->exact_rows, EX_MAX_ROWS)) t_single = (double)nq * (8.0 + kept_bytes / 3.5e6);
      else t_single = (double)nq * (25.0 + kept_bytes / 5.5e6);
      if (kern >= 2 && kept >= 1 && kept <= 16384 && kept * 24 <= s->rows)  // (listed mode's conditions, shard_search_batch)
        t_batch = 180.0 + (double)kept * (double)s->dim * 2.0 / 4.0e6 * ((nq + 127) / 128) + 0.3 * (double)nq;
    }
    enough = t_single > t_batch;
  }
  return enough && !s->safe_mode() && s->rows >= 4096 && k <= 1024 && s->rows < 0x7FFFFF00ll;
}

// nq queries on one shard: matrix-core batch when it pays, single-query pipeline
// otherwise and for whatever the batch hands back.
int shard_search_any(Shard *s, BatchCtx *b, int32_t batch_min_nq, const float *queries, int32_t nq, int32_t k,
                     const MaskSrc &mask, int32_t entries, SearchOut *out) {
  // (a mask's kept rows, for the cost model: a handle knows them; a pointer mask is counted when the answer can matter)
  int64_t kept = -1;
  if (mask && batch_min_nq == 1 && nq >= 2 && nq <= 512) kept = mask.part ? mask.part->kept : mask_kept_rows(s, mask.bytes);
  const bool use_batch = shard_takes_batch(s, batch_min_nq, nq, k, kept);
  if (!use_batch) return shard_search_blocks(s, queries, nq, k, mask, entries, out, PIPE_DEPTH);
  std::vector<int32_t> redo;
  const size_t bb = (size_t)tsh_candidate_block_bytes(entries);
  // the dense sample matrix (nq_pad x n_sample floats) is kept under 8 GB per call
  const int64_t per_q = batch_sample_rows(s->rows, k) * 4;
  const int32_t chunk = (int32_t)std::max<int64_t>(256, (int64_t)(8e9 / (double)per_q) / 256 * 256);
  for (int32_t q0 = 0; q0 < nq; q0 += chunk) {
    const int32_t nc = std::min(chunk, nq - q0);
    SearchOut part;
    part.h_blocks = out->h_blocks ? out->h_blocks + (size_t)q0 * bb : nullptr;
    part.d_blocks = out->d_blocks ? out->d_blocks + (size_t)q0 * bb : nullptr;
    part.user_stream = out->user_stream;
    part.extra = out->extra;
    part.q_base = out->q_base + q0;
    part.fin_thr = out->fin_thr;
    if (out->on_final)
      part.on_final = [out, q0, k](int32_t a, int32_t b2, const char *skip, const int64_t *ids, const double *dist, const int32_t *cnt) {
        out->on_final(q0 + a, q0 + b2, skip - q0, ids - (size_t)q0 * k, dist - (size_t)q0 * k, cnt - q0);
      };
    if (out->on_chunk)  // indices of the callback are the caller's: shift this part's
      part.on_chunk = [out, q0, bb](int32_t a, int32_t b2, const char *skip, const uint8_t *base, const double *mag_a) {
        out->on_chunk(q0 + a, q0 + b2, skip - q0, base - (size_t)q0 * bb, mag_a - q0);
      };
    std::vector<int32_t> r;
    const int force = s->planes_denied.load() > 0 && s->batch_kernel != 0 ? 0 : -1;
    int rc = shard_search_batch(s, b, queries + (size_t)q0 * s->dim, nc, k, mask, entries, &part, &r, force);
    // A device too full for this call's allocations is no reason to fail a search the rows themselves can answer:
    // first without the converted copy of the rows (f32 MFMA kernel on the rows as stored, smaller scratch), then
    // without the batched path at all (pipelined single-query scans need nothing beyond their contexts).  Same
    // results on every path; tsh_counters says which ran.
    if (rc == TSH_E_OOM && force != 0 && s->batch_kernel != 0) {
      r.clear();
      rc = shard_search_batch(s, b, queries + (size_t)q0 * s->dim, nc, k, mask, entries, &part, &r, 0);
      if (rc == TSH_OK) {
        s->c_plane_fallbacks++;
        s->planes_denied.store(64);  // the copy is tried again after that many calls, not on every one
      }
    } else if (rc == TSH_OK && force == 0) {
      s->c_plane_fallbacks++;
      s->planes_denied.fetch_sub(1);
    }
    if (rc == TSH_E_OOM) {
      s->c_scan_fallbacks++;
      r.clear();
      // (no callbacks: the blocks land in the caller's buffers and the caller finalises them, as it does for
      // queries the batch hands back)
      std::vector<std::vector<BlockEntry>> sp((size_t)(out->spill ? nc : 0)), ex((size_t)(out->extra ? nc : 0));
      SearchOut scans;
      scans.h_blocks = part.h_blocks;
      scans.d_blocks = part.d_blocks;
      scans.user_stream = part.user_stream;
      scans.spill = out->spill ? &sp : nullptr;
      scans.extra = out->extra ? &ex : nullptr;
      rc = shard_search_blocks(s, queries + (size_t)q0 * s->dim, nc, k, mask, entries, &scans, PIPE_DEPTH);
      for (int32_t q = 0; q < nc && rc == TSH_OK; ++q) {
        if (out->spill) (*out->spill)[(size_t)(q0 + q)] = std::move(sp[(size_t)q]);
        if (out->extra) (*out->extra)[(size_t)(out->q_base + q0 + q)] = std::move(ex[(size_t)q]);
      }
    }
    if (rc) return rc;
    for (int32_t q : r) redo.push_back(q0 + q);
  }
  int rc = TSH_OK;
  for (int32_t q : redo) {
    SearchOut one;
    std::vector<std::vector<BlockEntry>> sp(1), ex(1);
    one.h_blocks = out->h_blocks ? out->h_blocks + (size_t)q * bb : nullptr;
    one.d_blocks = out->d_blocks ? out->d_blocks + (size_t)q * bb : nullptr;
    one.user_stream = out->user_stream;
    one.spill = out->spill ? &sp : nullptr;
    one.extra = out->extra ? &ex : nullptr;
    rc = shard_search_blocks(s, queries + (size_t)q * s->dim, 1, k, mask, entries, &one, 1);
    if (rc) return rc;
    if (out->spill) (*out->spill)[(size_t)q] = std::move(sp[0]);
    if (out->extra) (*out->extra)[(size_t)(out->q_base + q)] = std::move(ex[0]);
  }
  return TSH_OK;
}

