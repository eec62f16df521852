// tsh_host_comm.inl.h -- the exchange of a row-sharded index, reachable from the C ABI (one process per GPU)
// Part of the single translation unit tsh_lib.hip (textually included there; not compiled alone).
//
// BASELINE.json's north star: "the corpus shards by row-range across the 8 GPUs of one node with a RCCL
// all-gather of per-shard (distance, row-id) top-k candidates over xGMI and a final host-side merge".  The Python
// harness can do that exchange with torch.distributed (tostore_amd/sharded.py); a Dart host has no torch, so the
// same steps are offered here behind plain C entry points.  librccl is loaded with dlopen on first use: a
// process that never shards does not pull it in, and a process that already holds a librccl (a torch host)
// gets that one.
//
// One tsh_search_sharded call, nq queries, W ranks:
//   * the queries are exchanged in GROUPS, but scanned as ONE pipeline: a progressive shard search
//     (tsh_search_shard_begin: a library thread keeps this rank's scans back to back over all queries of the call)
//     fills this rank's device blocks while the calling thread exchanges and merges every group whose blocks are
//     final -- group boundaries do not exist for the GPU (round 4 scanned group by group and paid the pipeline's
//     fill and drain, ~45 us, per group);
//   * exchange of a group: all-gather of every rank's candidate blocks (RCCL: device to device over xGMI);
//     rank r then copies back and merges only ITS SLICE of the group's queries (W blocks per query), so the host
//     work of a group is done once, spread over the ranks, not W times;
//   * a second, small all-gather carries every slice's final (ids, distances, counts) plus a status word per
//     rank, so every rank returns the full answer -- and the same verdict: a rank whose shard search failed
//     contributes blocks that say so (it does not leave the collective), a truncated block makes every rank retry
//     the group with the same larger entry count.
// Buffers grow on the same calls on every rank (same arguments everywhere, by contract); a call that grows them
// ends the growth with a tiny agreement all-gather, so an allocation failure on one rank fails the call on all.
// What cannot be recovered: a failing all-gather / stream.  The call then returns TSH_E_RCCL or TSH_E_HIP and the
// communicator must be destroyed (its peers may be blocked in the collective).
#include <dlfcn.h>

namespace {

struct RcclId {
  char internal[128];  // ncclUniqueId (NCCL_UNIQUE_ID_BYTES)
};
struct RcclApi {
  int (*GetUniqueId)(RcclId *) = nullptr;
  int (*CommInitRank)(void **, int, RcclId, int) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  std::string err;
  bool ok = false;
  bool overridden = false;  // TSH_RCCL_LIB named the library
};

RcclApi *rccl() {
  static RcclApi *api = [] {
    RcclApi *a = new RcclApi();
    void *h = nullptr;
    // TSH_RCCL_LIB: another library with the same five entry points.  tests/fake_rccl (several ranks on ONE GPU,
    // which the real library refuses) is the one user; a path that does not load is an error, not a reason to
    // fall back to the system's librccl behind the host's back.  Obeyed only in a process that switched the test
    // hooks on BEFORE its first tsh_comm_* call (TSH_OPT_TEST_HOOKS): a database's environment must not be able to
    // make it dlopen a path
    if (const char *over = test_env("TSH_RCCL_LIB"); over && *over) {
      h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
      if (!h) {
        const char *why = dlerror();
        a->err = std::string("TSH_RCCL_LIB=") + over + " does not load: " + (why ? why : "");
        return a;
      }
      a->overridden = true;
    }
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      if (h) break;
      h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    }
    if (!h) {
      const char *why = dlerror();
      a->err = std::string("librccl not found: ") + (why ? why : "");
      return a;
    }
    auto sym = [&](const char *n) -> void * {
      void *p = dlsym(h, n);
      if (!p && a->err.empty()) a->err = std::string("librccl lacks ") + n;
      return p;
    };
    a->GetUniqueId = reinterpret_cast<decltype(a->GetUniqueId)>(sym("ncclGetUniqueId"));
    a->CommInitRank = reinterpret_cast<decltype(a->CommInitRank)>(sym("ncclCommInitRank"));
    a->AllGather = reinterpret_cast<decltype(a->AllGather)>(sym("ncclAllGather"));
    a->CommDestroy = reinterpret_cast<decltype(a->CommDestroy)>(sym("ncclCommDestroy"));
    a->GetErrorString = reinterpret_cast<decltype(a->GetErrorString)>(sym("ncclGetErrorString"));
    a->ok = a->err.empty();
    return a;
  }();
  return api;
}

int rccl_fail(const char *what, int rc) {
  RcclApi *r = rccl();
  return set_err(TSH_E_RCCL, "%s failed: %s", what, r->GetErrorString ? r->GetErrorString(rc) : "?");
}

// (sharded_group_max / sharded_schedule: tsh_host_sync.h -- pure functions, tested on the CPU)
// header of one rank's result slice in the second all-gather
struct ResHeader {
  int32_t status;  // TSH_OK, or why this rank could not merge its slice (its own error, or TSH_E_PEER)
  int32_t need;    // entries a truncated block of the slice asks for (0 = none)
  int32_t pad[14];
};
static_assert(sizeof(ResHeader) == 64, "result header is 64 bytes");
constexpr uint32_t FLAG_RANK_ERROR = 0x80000000u;  // BlockHeader.flags: this rank's shard search failed (pad[0] = rc)

inline size_t res_rec_bytes(int32_t k) { return (size_t)k * 16 + 8; }  // k ids, k distances, count + pad

// Small groups are merged WHOLE on every rank: each rank copies back all W x gq gathered blocks and finalises all gq
// queries itself, and the second all-gather (every slice's results to every rank: an H2D, a collective, a D2H and a
// wait -- the larger half of a small group's exchange) does not happen.  W times the merge work, which for a few
// queries is microseconds; every rank reads the same gathered bytes, so every rank reaches the same verdict without
// being told.  Matters where the exchange is exposed: the last group of a call, i.e. short calls on many GPUs (the
// driver's 20-query regions at N = 8 are three groups).  Larger groups keep the per-rank slices.
constexpr size_t WHOLE_MERGE_MAX_BLOCKS = 128;  // W x gq
inline bool merge_whole(size_t W, size_t gq) { return W > 1 && W * gq <= WHOLE_MERGE_MAX_BLOCKS; }
// queries a rank may have to merge in a group of up to gq: its slice, or a whole small group
inline size_t merge_queries_max(size_t W, size_t gq) {
  const size_t slice = (gq + W - 1) / W, whole = W > 1 ? std::min(gq, WHOLE_MERGE_MAX_BLOCKS / W) : 0;
  return std::max(slice, whole);
}

}  // namespace

struct tsh_comm {
  void *comm = nullptr;  // ncclComm_t (RCCL transport)
  tsh_allgather_fn host_fn = nullptr;  // host transport (tsh_comm_create_host)
  void *host_user = nullptr;
  int32_t world = 1, rank = 0, device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev = nullptr;
  std::mutex mu;  // collectives of one communicator are issued one call at a time
  uint8_t *d_mine = nullptr;  // this rank's blocks of a whole call (or of a window of SHARDED_WINDOW queries of it)
  size_t mine_cap = 0;
  uint8_t *d_retry = nullptr;  // this rank's blocks of a group redone with larger blocks (ties)
  size_t retry_cap = 0;
  uint8_t *h_mine = nullptr;  // host transport: pinned copy of d_mine
  size_t h_mine_cap = 0;
  uint8_t *d_all = nullptr;  // RCCL: every rank's blocks of the group
  size_t all_cap = 0;
  uint8_t *h_slice = nullptr;  // pinned: RCCL -- W x (this rank's query slice) blocks; host transport -- W x group blocks
  size_t slice_cap = 0;
  uint8_t *h_res_mine = nullptr, *h_res_all = nullptr, *d_res_mine = nullptr, *d_res_all = nullptr;
  size_t res_cap = 0;  // bytes of one rank's result slice
  int64_t *d_agree = nullptr, *h_agree = nullptr;  // (1 + world) x {status, rows of the rank's shard} (allocated with the communicator)
  int64_t scan_bytes_hint = 0;  // bytes one query's scan reads on the largest shard of any rank, as of the last agreement:
                                // the same number on every rank (a rank's own handle may be bad: nothing rank-local may
                                // decide how a call is cut into groups)
  uint32_t tag_seq = 0;  // generation of this rank's blocks: one per window of a call (the same on every rank)
  uint64_t exchanges = 0;  // block all-gathers enqueued so far
  bool timed_now = false;  // the one in flight carries ev_t[0] / ev_t[1]
  int64_t tl_exch = 0, tl_timed = 0;  // exchanges / timed exchanges since the timeline's last reset: gather_us and
                                      // slice_d2h_us are sums over the TIMED ones, scaled when the timeline is read
  int32_t calls_since_agree = 0;      // calls since the ranks last told each other their shard sizes
  bool batch_hint = false;            // some rank's handle sends calls to the matrix cores (TSH_OPT_BATCH_MIN_NQ != 0), as of
                                      // the last agreement: rides in the low bit of the bytes figure (a multiple of four)
  int32_t group = 0;  // queries per exchange; 0 = by the size of the call
  std::unique_ptr<OneWorker> worker;  // runs the calls' progressive shard searches (one call at a time: mu)
  hipEvent_t ev_t[3] = {nullptr, nullptr, nullptr};  // RCCL: before / after the block all-gather, after the slice's D2H
  tsh_comm_timeline tl = {};  // guarded by mu
  std::atomic<int64_t> scan_ns{0};  // the scanning thread's busy time (+ retries' scans)
};

namespace {

int comm_sync(tsh_comm *c) {  // the communicator's stream (an event: blocking where CPUs are scarce, see comm_common_create)
  HIPCHK(hipEventRecord(c->ev, c->stream));
  HIPCHK(hipEventSynchronize(c->ev));
  return TSH_OK;
}

// all-gather of `bytes` per rank.  RCCL: device buffers on the communicator's stream (asynchronous);
// host transport: host buffers, synchronous
int comm_allgather_dev(tsh_comm *c, const void *d_send, void *d_recv, size_t bytes) {
  int nrc = rccl()->AllGather(d_send, d_recv, bytes, /*ncclChar*/ 0, c->comm, c->stream);
  if (nrc != 0) return rccl_fail("ncclAllGather", nrc);
  return TSH_OK;
}
int comm_allgather_host(tsh_comm *c, const void *h_send, void *h_recv, size_t bytes) {
  int32_t rc = c->host_fn(c->host_user, h_send, h_recv, (int64_t)bytes);
  if (rc != 0) return set_err(TSH_E_RCCL, "the host's all-gather callback failed (%d)", rc);
  return TSH_OK;
}

// every rank says whether its part of a step that may fail locally (allocations) worked; all ranks get the
// same answer.  Collective.  The same exchange tells every rank how many bytes a scan of the largest shard reads
// (scan_bytes_hint: what the group schedule of later calls is sized by -- identical on every rank, because it only
// changes here).
int comm_agree(tsh_comm *c, int local_rc, int64_t local_rows) {
  if (c->world == 1) {
    c->scan_bytes_hint = local_rows & ~(int64_t)3;
    c->batch_hint = (local_rows & 1) != 0;
    return local_rc;
  }
  c->h_agree[0] = local_rc;
  c->h_agree[1] = local_rows;
  int64_t *all = c->h_agree + 2;
  if (c->host_fn) {
    int rc = comm_allgather_host(c, c->h_agree, all, 16);
    if (rc) return rc;
  } else {
    HIPCHK(hipMemcpyAsync(c->d_agree, c->h_agree, 16, hipMemcpyHostToDevice, c->stream));
    int rc = comm_allgather_dev(c, c->d_agree, c->d_agree + 2, 16);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(all, c->d_agree + 2, 16 * (size_t)c->world, hipMemcpyDeviceToHost, c->stream));
    rc = comm_sync(c);
    if (rc) return rc;
  }
  int64_t rows = 0;
  bool batches = false;
  for (int r = 0; r < c->world; ++r) {
    rows = std::max<int64_t>(rows, all[2 * r + 1] & ~(int64_t)3);
    batches = batches || (all[2 * r + 1] & 1) != 0;
  }
  c->scan_bytes_hint = rows;
  c->batch_hint = batches;
  if (local_rc != TSH_OK) return local_rc;
  for (int r = 0; r < c->world; ++r)
    if (all[2 * r] != TSH_OK)
      return set_err(TSH_E_PEER, "rank %d could not allocate its exchange buffers (%d)", r, (int)all[2 * r]);
  return TSH_OK;
}

template <typename T>
int grow_dev(T **p, size_t *cap, size_t want) {
  if (want <= *cap) return TSH_OK;
  hipFree(*p);
  *p = nullptr;
  *cap = 0;
  HIPCHK(hipMalloc(reinterpret_cast<void **>(p), want));
  *cap = want;
  return TSH_OK;
}
template <typename T>
int grow_host(T **p, size_t *cap, size_t want) {
  if (want <= *cap) return TSH_OK;
  hipHostFree(*p);
  *p = nullptr;
  *cap = 0;
  HIPCHK(hipHostMalloc(reinterpret_cast<void **>(p), want, hipHostMallocDefault));
  *cap = want;
  return TSH_OK;
}

// buffers for groups of up to gq queries with `entries` entries per block; this rank's own blocks: call_q queries'
// worth (the window of the call the scans run ahead in), or -- retry -- one group's in d_retry.  *grew: anything was
// (re)allocated -- identical on every rank, since capacities only depend on the calls made so far
int comm_reserve(tsh_comm *c, int32_t gq, int32_t entries, int32_t k, int32_t call_q, bool retry, bool *grew) {
  const size_t bb = (size_t)tsh_candidate_block_bytes(entries), W = (size_t)c->world;
  const size_t slice_q = merge_queries_max(W, (size_t)gq);
  const size_t mine = bb * (size_t)gq, res = sizeof(ResHeader) + slice_q * res_rec_bytes(k);
  const size_t slice = c->host_fn ? W * mine : W * slice_q * bb;
  const size_t own = retry ? 0 : bb * (size_t)call_q, again = retry ? mine : 0;
  *grew = own > c->mine_cap || again > c->retry_cap || res > c->res_cap || slice > c->slice_cap ||
          (c->host_fn ? mine > c->h_mine_cap : W * mine > c->all_cap);
  if (!*grew) return TSH_OK;
  int rc = grow_dev(&c->d_mine, &c->mine_cap, own);
  if (!rc) rc = grow_dev(&c->d_retry, &c->retry_cap, again);
  if (!rc) rc = grow_host(&c->h_slice, &c->slice_cap, slice);
  if (!rc && c->host_fn) rc = grow_host(&c->h_mine, &c->h_mine_cap, mine);
  if (!rc && !c->host_fn) rc = grow_dev(&c->d_all, &c->all_cap, W * mine);
  if (!rc && res > c->res_cap) {
    hipHostFree(c->h_res_mine);
    hipHostFree(c->h_res_all);
    hipFree(c->d_res_mine);
    hipFree(c->d_res_all);
    c->h_res_mine = c->h_res_all = c->d_res_mine = c->d_res_all = nullptr;
    c->res_cap = 0;
    size_t cap = 0;
    rc = grow_host(&c->h_res_mine, &cap, res);
    cap = 0;
    if (!rc) rc = grow_host(&c->h_res_all, &cap, W * res);
    if (!rc && !c->host_fn) {
      cap = 0;
      rc = grow_dev(&c->d_res_mine, &cap, res);
      cap = 0;
      if (!rc) rc = grow_dev(&c->d_res_all, &cap, W * res);
    }
    if (!rc) c->res_cap = res;
  }
  return rc;
}

// after a growth some rank could not follow, every rank drops its buffers: capacities must stay identical on all
// ranks, or the next call's "did anything grow" would differ from rank to rank
void comm_drop_buffers(tsh_comm *c) {
  hipFree(c->d_mine);
  hipFree(c->d_retry);
  c->d_mine = c->d_retry = nullptr;
  c->mine_cap = c->retry_cap = 0;
  hipHostFree(c->h_mine);
  hipFree(c->d_all);
  hipHostFree(c->h_slice);
  hipHostFree(c->h_res_mine);
  hipHostFree(c->h_res_all);
  hipFree(c->d_res_mine);
  hipFree(c->d_res_all);
  c->h_mine = c->d_all = c->h_slice = c->h_res_mine = c->h_res_all = c->d_res_mine = c->d_res_all = nullptr;
  c->h_mine_cap = c->all_cap = c->slice_cap = c->res_cap = 0;
}
// refresh: agree although nothing grew (the ranks' shard sizes, which size the group schedule, drift with appends:
// tsh_search_sharded asks for it every 64th call -- the same calls on every rank, the call being collective)
int comm_reserve_agreed(tsh_comm *c, int32_t gq, int32_t entries, int32_t k, int32_t call_q, bool retry, int64_t rows,
                        bool *grew, bool refresh = false) {
  int rc = comm_reserve(c, gq, entries, k, call_q, retry, grew);
  if (*grew || refresh) {
    rc = comm_agree(c, rc, rows);
    if (rc) comm_drop_buffers(c);
  }
  return rc;
}

// this rank's blocks of a group, as error markers (its shard search failed with `rc`): the rank stays in the collective
int comm_error_blocks(uint8_t *d_blocks, int32_t gq, int32_t entries, int rc_local) {
  const size_t bb = (size_t)tsh_candidate_block_bytes(entries);
  std::vector<BlockHeader> h((size_t)gq);
  memset(h.data(), 0, h.size() * sizeof(BlockHeader));
  for (auto &x : h) {
    x.entries = (uint32_t)entries;
    x.flags = FLAG_RANK_ERROR;
    x.pad[0] = (uint32_t)rc_local;
  }
  HIPCHK(hipMemcpy2D(d_blocks, bb, h.data(), sizeof(BlockHeader), sizeof(BlockHeader), (size_t)gq,
                     hipMemcpyHostToDevice));
  return TSH_OK;
}

struct GroupOut {
  int32_t need = 0;  // > 0: every rank retries the group with this many entries
};

// RCCL transport: the device side of a group's exchange on the communicator's stream -- [wait for the events behind
// the group's last block writers |] all-gather of every rank's blocks | this rank's query slice of them to the host |
// event.  With events the calls are made AHEAD, while the group's scans still run: launching a collective costs the
// host tens of microseconds (RCCL's enqueue path), which for the LAST group of a call sat between the last block
// becoming final and the all-gather starting; stream-ordered, the all-gather starts a few microseconds after the
// last re-rank whatever the host is doing.  What the device cannot know is whether the HOST will accept a block as
// final (ties that overflow a candidate list are redone by a wider pass the host starts): such a block carries
// FLAG_LIST_OVERFLOW when it is gathered, which every rank that reads it answers like a truncated block -- the group is
// redone with larger blocks.  And a rank whose scans fail half-way leaves blocks of an EARLIER call in place: every
// block carries the generation it was written for (BlockHeader.pad[1]), and a block of another generation is a failed
// peer, not an answer.
int comm_exchange_enqueue(tsh_comm *c, const uint8_t *d_blocks, int32_t gq, int32_t entries, const hipEvent_t *after,
                          int n_after) {
  const size_t bb = (size_t)tsh_candidate_block_bytes(entries), W = (size_t)c->world, mine = bb * (size_t)gq;
  const bool whole = merge_whole(W, (size_t)gq);
  const int32_t slice_q = whole ? gq : (int32_t)(((size_t)gq + W - 1) / W);
  const int32_t a = whole ? 0 : std::min<int32_t>(gq, c->rank * slice_q), b = std::min<int32_t>(gq, a + slice_q);
  for (int i = 0; i < n_after; ++i) HIPCHK(hipStreamWaitEvent(c->stream, after[i], 0));
  // (the split of the device time into all-gather and copy is sampled, every fourth exchange: each record is a packet
  // the queue works through, a few microseconds that the last exchange of a call shows)
  c->timed_now = (c->exchanges++ & 3) == 0;
  if (c->timed_now) HIPCHK(hipEventRecord(c->ev_t[0], c->stream));
  int rc = comm_allgather_dev(c, d_blocks, c->d_all, mine);  // k' x 24 B per rank and query: latency-bound
  if (rc) return rc;
  if (c->timed_now) HIPCHK(hipEventRecord(c->ev_t[1], c->stream));
  // every query of the group (a world of one, or a small group merged whole) is one contiguous copy; the pitched
  // form took ~30 us of device time for 60 KB
  if (b - a == gq)
    HIPCHK(hipMemcpyAsync(c->h_slice, c->d_all, W * mine, hipMemcpyDeviceToHost, c->stream));
  else if (b > a)
    HIPCHK(hipMemcpy2DAsync(c->h_slice, (size_t)(b - a) * bb, c->d_all + (size_t)a * bb, mine, (size_t)(b - a) * bb, W,
                            hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipEventRecord(c->ev_t[2], c->stream));
  return TSH_OK;
}

// Exchange + merge of one group whose blocks sit at d_blocks (device).  local_rc: what this rank's shard search said.
// enqueued: comm_exchange_enqueue has run for this group already (stream-ordered behind the group's block writers);
// then a rank whose scans failed cannot mark its blocks any more -- its peers see that from the blocks' generation
// (tag: what BlockHeader.pad[1] of every block of this group must say; 0 = blocks of a synchronous shard search).
// Collective; returns the same verdict class on every rank (own error / TSH_E_PEER / TSH_OK + need).
int comm_exchange_group(tsh_comm *c, tsh_index *shard, uint8_t *d_blocks, int local_rc, const float *queries, int32_t gq,
                        int32_t k, double thr, int32_t entries, int64_t *out_ids, double *out_dist, int32_t *out_count,
                        GroupOut *go, bool enqueued = false, uint32_t tag = 0) {
  const size_t bb = (size_t)tsh_candidate_block_bytes(entries), W = (size_t)c->world, mine = bb * (size_t)gq;
  const bool whole = merge_whole(W, (size_t)gq);  // (same on every rank: W and gq are)
  const int32_t slice_q = whole ? gq : (int32_t)(((size_t)gq + W - 1) / W);
  const int32_t a = whole ? 0 : std::min<int32_t>(gq, c->rank * slice_q), b = std::min<int32_t>(gq, a + slice_q);
  const size_t rec = res_rec_bytes(k), res = sizeof(ResHeader) + (size_t)slice_q * rec;
  int rc;
  tsh_comm_timeline &tl = c->tl;
  if (local_rc != TSH_OK && !enqueued) {
    rc = comm_error_blocks(d_blocks, gq, entries, local_rc);
    if (rc) return rc;
  }
  // ---- 1. every rank's blocks of the group; this rank's query slice of them to the host -------------------
  const uint8_t *slice_base;  // block (rank w, query a + i) at slice_base + w * slice_pitch + i * bb
  size_t slice_pitch;
  const double t1 = now_us();
  if (c->host_fn) {
    HIPCHK(hipMemcpyAsync(c->h_mine, d_blocks, mine, hipMemcpyDeviceToHost, c->stream));
    rc = comm_sync(c);
    if (rc) return rc;
    const double t1b = now_us();
    rc = comm_allgather_host(c, c->h_mine, c->h_slice, mine);
    if (rc) return rc;
    tl.slice_d2h_us += t1b - t1;  // (host transport: the rank's whole group goes to the host, not a slice)
    tl.gather_us += now_us() - t1b;
    c->tl_exch++;
    c->tl_timed++;
    slice_base = c->h_slice + (size_t)a * bb;
    slice_pitch = mine;
  } else {
    if (!enqueued) {
      rc = comm_exchange_enqueue(c, d_blocks, gq, entries, nullptr, 0);
      if (rc) return rc;
    }
    HIPCHK(hipEventSynchronize(c->ev_t[2]));
    c->tl_exch++;
    if (c->timed_now) {
      float ms_g = 0.f, ms_d = 0.f;  // device time on the communicator's stream; the all-gather's includes the wait
      HIPCHK(hipEventElapsedTime(&ms_g, c->ev_t[0], c->ev_t[1]));  // for the slowest rank to arrive
      HIPCHK(hipEventElapsedTime(&ms_d, c->ev_t[1], c->ev_t[2]));
      tl.gather_us += 1e3 * ms_g;  // (one exchange in four is timed: tsh_comm_get_timeline scales the sums by
      tl.slice_d2h_us += 1e3 * ms_d;  // exchanges / timed exchanges, so a run of one or two exchanges reports what it saw)
      c->tl_timed++;
    }

    slice_base = c->h_slice;
    slice_pitch = (size_t)(b - a) * bb;
  }
  const double t2 = now_us();
  tl.exchange_wait_us += t2 - t1;
  // ---- 2. merge the slice ---------------------------------------------------------------------------------
  ResHeader *rh = reinterpret_cast<ResHeader *>(c->h_res_mine);
  memset(rh, 0, sizeof *rh);
  rh->status = local_rc;
  uint8_t *recs = c->h_res_mine + sizeof(ResHeader);
  if (local_rc == TSH_OK && b > a) {
    uint32_t need = 0;
    for (size_t w = 0; w < W && rh->status == TSH_OK; ++w)
      for (int32_t i = 0; i < b - a; ++i) {
        const BlockHeader *h = reinterpret_cast<const BlockHeader *>(slice_base + w * slice_pitch + (size_t)i * bb);
        if (h->flags & FLAG_RANK_ERROR) {
          rh->status = TSH_E_PEER;
          rh->pad[0] = (int32_t)w;
          rh->pad[1] = (int32_t)h->pad[0];
          break;
        }
        if (h->pad[1] != tag) {  // a block of another call: rank w's scans did not get this far
          rh->status = TSH_E_PEER;
          rh->pad[0] = (int32_t)w;
          rh->pad[1] = TSH_E_HIP;
          break;
        }
        if (h->entries != (uint32_t)entries) {
          rh->status = TSH_E_FORMAT;
          break;
        }
        if (h->count > h->entries) need = std::max(need, h->count);
        // (gathered stream-ordered, before its rank's host had a look: the wide pass is still to come)
        else if (h->flags & FLAG_LIST_OVERFLOW) need = std::max(need, h->entries + 64u);
      }
    if (rh->status == TSH_OK && need) rh->need = (int32_t)round_up(need, 64);
    if (rh->status == TSH_OK && !need) {
      const int metric = shard->metric, dim = shard->dim;
      parallel_for(b - a, [&](int32_t i) {
        EntryList lists[64];
        std::vector<EntryList> big;
        EntryList *lp = lists;
        if (W > 64) {
          big.resize(W);
          lp = big.data();
        }
        for (size_t w = 0; w < W; ++w) {
          const uint8_t *p = slice_base + w * slice_pitch + (size_t)i * bb;
          lp[w] = {reinterpret_cast<const BlockEntry *>(p + sizeof(BlockHeader)),
                   reinterpret_cast<const BlockHeader *>(p)->count};
        }
        uint8_t *r = recs + (size_t)i * rec;
        int64_t *ids = reinterpret_cast<int64_t *>(r);
        double *dd = reinterpret_cast<double *>(r + (size_t)k * 8);
        int32_t *cnt = reinterpret_cast<int32_t *>(r + (size_t)k * 16);
        cnt[0] = finalize_query(metric, dim, queries + (size_t)(a + i) * dim, k, thr, lp, W, ids, dd);
        cnt[1] = 0;
      });
    }
  }
  // ---- 3. every slice's results + status to every rank ----------------------------------------------------
  const double t3 = now_us();
  tl.merge_us += t3 - t2;
  const uint8_t *all_res = c->h_res_mine;
  const size_t RW = whole ? 1 : W;  // result slices to read below: one (this rank's own, the whole group) or every rank's
  if (W > 1 && !whole) {
    if (c->host_fn) {
      rc = comm_allgather_host(c, c->h_res_mine, c->h_res_all, res);
      if (rc) return rc;
    } else {
      HIPCHK(hipMemcpyAsync(c->d_res_mine, c->h_res_mine, res, hipMemcpyHostToDevice, c->stream));
      rc = comm_allgather_dev(c, c->d_res_mine, c->d_res_all, res);
      if (rc) return rc;
      HIPCHK(hipMemcpyAsync(c->h_res_all, c->d_res_all, W * res, hipMemcpyDeviceToHost, c->stream));
      rc = comm_sync(c);
      if (rc) return rc;
    }
    all_res = c->h_res_all;
  }
  const double t4 = now_us();
  tl.result_gather_us += t4 - t3;
  int32_t need = 0;
  for (size_t w = 0; w < RW; ++w) {
    const ResHeader *h = reinterpret_cast<const ResHeader *>(all_res + w * res);
    if (h->status != TSH_OK) {
      if (local_rc != TSH_OK) return local_rc;  // (this rank's own error text is still in place)
      if (h->status == TSH_E_PEER)
        return set_err(TSH_E_PEER, "rank %d's shard search failed (%d); this call is void on every rank", h->pad[0], h->pad[1]);
      return set_err(h->status == TSH_E_FORMAT ? TSH_E_FORMAT : TSH_E_PEER, "rank %d could not merge its slice (%d)", (int)w,
                     h->status);
    }
    need = std::max(need, h->need);
  }
  go->need = need;
  if (need) return TSH_OK;
  for (size_t w = 0; w < RW; ++w) {
    const int32_t wa = std::min<int32_t>(gq, (int32_t)w * slice_q), wb = std::min<int32_t>(gq, wa + slice_q);
    const uint8_t *r = all_res + w * res + sizeof(ResHeader);
    for (int32_t q = wa; q < wb; ++q, r += rec) {
      memcpy(out_ids + (size_t)q * k, r, (size_t)k * 8);
      memcpy(out_dist + (size_t)q * k, r + (size_t)k * 8, (size_t)k * 8);
      out_count[q] = *reinterpret_cast<const int32_t *>(r + (size_t)k * 16);
    }
  }
  tl.copy_out_us += now_us() - t4;
  return TSH_OK;
}

int comm_common_create(tsh_comm *c, int32_t world, int32_t rank, int32_t device) {
  c->world = world;
  c->rank = rank;
  c->device = device;
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  // (waits spin unless the ranks outnumber the CPUs, blocking_wait(): an interrupt-driven wake-up costs tens of
  // microseconds, and the last exchange of every call is waited for in full view)
  const unsigned wait_flag = blocking_wait() ? hipEventBlockingSync : 0u;
  HIPCHK(hipEventCreateWithFlags(&c->ev, hipEventDisableTiming | wait_flag));
  for (auto &e : c->ev_t) HIPCHK(hipEventCreateWithFlags(&e, wait_flag));  // timed: the exchange's phases
  c->worker.reset(new OneWorker(blocking_wait() ? 0.0 : 300.0));  // (CPUs to spare: it polls 0.3 ms for the next call)
  HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_agree), 16 * (size_t)(world + 1)));
  HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&c->h_agree), 16 * (size_t)(world + 1), hipHostMallocDefault));
  return TSH_OK;
}

void comm_free(tsh_comm *c) {
  c->worker.reset();
  hipSetDevice(c->device);
  if (c->comm) rccl()->CommDestroy(c->comm);
  hipFree(c->d_mine);
  hipFree(c->d_retry);
  hipHostFree(c->h_mine);
  hipFree(c->d_all);
  hipHostFree(c->h_slice);
  hipHostFree(c->h_res_mine);
  hipHostFree(c->h_res_all);
  hipFree(c->d_res_mine);
  hipFree(c->d_res_all);
  hipFree(c->d_agree);
  hipHostFree(c->h_agree);
  if (c->ev) hipEventDestroy(c->ev);
  for (auto e : c->ev_t)
    if (e) hipEventDestroy(e);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}

int comm_check_create(const void *id_or_fn, int32_t world, int32_t rank, int32_t *device, tsh_comm **out) {
  if (!out) return set_err(TSH_E_BAD_ARG, "out is NULL");
  *out = nullptr;
  if (!id_or_fn || world < 1 || rank < 0 || rank >= world) return set_err(TSH_E_BAD_ARG, "bad id / callback / world / rank");
  if (device_count_cached() <= 0) return set_err(TSH_E_NO_DEVICE, "no HIP device available");
  if (*device < 0 && hipGetDevice(device) != hipSuccess) *device = 0;
  if (*device >= device_count_cached()) return set_err(TSH_E_BAD_ARG, "device %d not present", *device);
  HIPCHK(hipSetDevice(*device));
  return TSH_OK;
}

}  // namespace

extern "C" {

int32_t tsh_comm_unique_id(void *out_id) {
  if (!out_id) return set_err(TSH_E_BAD_ARG, "out_id is NULL");
  RcclApi *r = rccl();
  if (!r->ok) return set_err(TSH_E_RCCL, "%s", r->err.c_str());
  if (device_count_cached() <= 0) return set_err(TSH_E_NO_DEVICE, "no HIP device available");
  RcclId id;
  int rc = r->GetUniqueId(&id);
  if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
  memcpy(out_id, &id, sizeof id);
  return TSH_OK;
}

int32_t tsh_comm_create(const void *id, int32_t world, int32_t rank, int32_t device, tsh_comm **out) {
  int rc = comm_check_create(id, world, rank, &device, out);
  if (rc) return rc;
  RcclApi *r = rccl();
  if (!r->ok) return set_err(TSH_E_RCCL, "%s", r->err.c_str());
  tsh_comm *c = new tsh_comm();
  rc = comm_common_create(c, world, rank, device);
  if (rc) {
    comm_free(c);
    return rc;
  }
  RcclId uid;
  memcpy(&uid, id, sizeof uid);
  int nrc = r->CommInitRank(&c->comm, world, uid, rank);  // collective: every rank of the job calls it
  if (nrc != 0) {
    c->comm = nullptr;
    comm_free(c);
    return rccl_fail("ncclCommInitRank", nrc);
  }
  *out = c;
  return TSH_OK;
}

int32_t tsh_comm_create_host(int32_t world, int32_t rank, int32_t device, tsh_allgather_fn allgather, void *user,
                             tsh_comm **out) {
  int rc = comm_check_create(reinterpret_cast<const void *>(allgather), world, rank, &device, out);
  if (rc) return rc;
  tsh_comm *c = new tsh_comm();
  c->host_fn = allgather;
  c->host_user = user;
  rc = comm_common_create(c, world, rank, device);
  if (rc) {
    comm_free(c);
    return rc;
  }
  *out = c;
  return TSH_OK;
}

int32_t tsh_comm_destroy(tsh_comm *c) {
  if (!c) return TSH_OK;
  comm_free(c);
  return TSH_OK;
}

int32_t tsh_comm_world(tsh_comm *c) { return c ? c->world : 0; }

int32_t tsh_comm_set_group(tsh_comm *c, int32_t queries_per_exchange) {
  if (!c || queries_per_exchange < 0 || queries_per_exchange > 65536) return set_err(TSH_E_BAD_ARG, "bad comm / group");
  std::lock_guard<std::mutex> lk(c->mu);
  c->group = queries_per_exchange;
  return TSH_OK;
}

int32_t tsh_comm_get_timeline(tsh_comm *c, tsh_comm_timeline *out, int32_t reset) {
  if (!c || !out) return set_err(TSH_E_BAD_ARG, "comm / out is NULL");
  std::lock_guard<std::mutex> lk(c->mu);  // never in the middle of a call
  *out = c->tl;
  if (c->tl_timed > 0) {  // the sampled device-side split, scaled to all exchanges
    const double scale = (double)c->tl_exch / (double)c->tl_timed;
    out->gather_us *= scale;
    out->slice_d2h_us *= scale;
  }
  out->scan_us = 1e-3 * (double)c->scan_ns.load(std::memory_order_relaxed);
  out->world = c->world;
  out->rank = c->rank;
  out->transport = c->host_fn ? 1 : (rccl()->overridden ? 2 : 0);
  if (reset) {
    c->tl = tsh_comm_timeline{};
    c->tl_exch = c->tl_timed = 0;
    c->scan_ns.store(0, std::memory_order_relaxed);
  }
  return TSH_OK;
}

int32_t tsh_search_sharded(tsh_index *shard, tsh_comm *c, const float *queries, int32_t nq, int32_t k, double thr,
                           const uint8_t *row_mask, int64_t *out_ids, double *out_dist, int32_t *out_count) {
  // arguments that are the same on every rank by contract are answered locally ...
  if (!c) return set_err(TSH_E_BAD_ARG, "comm is NULL");
  if (nq < 0) return set_err(TSH_E_BAD_ARG, "nq < 0");
  if (nq == 0) return TSH_OK;
  if (!queries || !out_count) return set_err(TSH_E_BAD_ARG, "queries / out_count is NULL");
  for (int32_t q = 0; q < nq; ++q) out_count[q] = 0;
  if (k <= 0) return TSH_OK;
  if (k > (1 << 20)) return set_err(TSH_E_BAD_ARG, "k too large");
  if (!out_ids || !out_dist) return set_err(TSH_E_BAD_ARG, "out_ids / out_dist is NULL");
  // ... what can differ from rank to rank (the handle, its device, its search) travels through the exchange
  int local_rc = TSH_OK;
  if (!shard || shard->shards.size() != 1)
    local_rc = set_err(TSH_E_BAD_ARG, "needs a single-shard handle (tsh_index_create_shard)");
  else if (shard->shards[0]->device != c->device)
    local_rc = set_err(TSH_E_BAD_ARG, "shard and communicator sit on different devices");
  std::string local_err = g_err;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(hipSetDevice(c->device));
  const int32_t entries = tsh_default_block_entries(k);
  // buffers: for the largest group any schedule of this call may hold, and this rank's own blocks for a window of the
  // call (whole groups; 4096 queries = 25 MB at k = 100)
  const int32_t Gmax = c->group > 0 ? std::min(c->group, nq) : sharded_group_max(nq);
  const int32_t win_cap = std::min(nq, std::max(Gmax, SHARDED_WINDOW));
  int64_t my_rows = 0;  // (bytes one scan of this rank's shard reads: what the ranks tell each other)
  if (local_rc == TSH_OK) {
    Shard *s0 = shard->shards[0].get();
    std::shared_lock<RwLock> sl = share(shard, s0);
    my_rows = s0->rows * s0->ld * 4;
    if (shard->batch_min_nq != 0) my_rows |= 1;  // (this rank batches: the schedule's other input, in the figure's spare low bit)
  }
  bool grew = false;
  const double t_in = now_us();
  tsh_comm_timeline &tl = c->tl;
  struct CallClock {  // every way out of the call books its wall time
    tsh_comm_timeline &tl;
    double t0;
    int32_t nq;
    ~CallClock() {
      tl.calls++;
      tl.queries += nq;
      tl.call_us += now_us() - t0;
    }
  } call_clock{tl, t_in, nq};
  // (scan_bytes_hint sizes the schedule below: in a world of one comm_agree just takes this rank's figure; with peers
  // it is refreshed whenever the buffers grow and on every 64th call, so it follows shards that grow or shrink)
  const bool refresh = c->world == 1 || ++c->calls_since_agree >= 64;
  if (refresh) c->calls_since_agree = 0;
  int rc = comm_reserve_agreed(c, Gmax, entries, k, win_cap, false, my_rows, &grew, refresh);
  tl.reserve_us += now_us() - t_in;
  if (rc) return rc;
  const int32_t dim = local_rc == TSH_OK ? shard->dim : 0;
  // queries per exchange, group after group (the same list on every rank: nq, the communicator's setting and the
  // rows the ranks told each other are all it depends on)
  std::vector<int32_t> sizes;
  if (c->group > 0) {
    for (int32_t q = 0; q < nq; q += Gmax) sizes.push_back(std::min(Gmax, nq - q));
  } else {
    sharded_schedule(nq, (double)c->scan_bytes_hint / 6.5e6, &sizes, c->batch_hint);
  }
  const size_t bb = (size_t)tsh_candidate_block_bytes(entries);
  size_t gi = 0;  // next group
  for (int32_t w0 = 0; w0 < nq;) {
    // a window: as many whole groups as fit this rank's block buffer (at least one)
    size_t ge = gi;
    int32_t wn = 0;
    while (ge < sizes.size() && (ge == gi || wn + sizes[ge] <= win_cap)) wn += sizes[ge++];
    // this rank's scans of the window: one pipeline on a library thread, blocks final in query order
    tsh_shard_stream *ss = nullptr;
    int scan_rc = local_rc;
    std::string scan_err = local_err;
    // the generation this window's blocks carry in BlockHeader.pad[1] (never 0; the same on every rank, whatever a
    // rank's own state: the counter moves with the windows, and those are the same everywhere)
    if (++c->tag_seq == 0) ++c->tag_seq;
    const uint32_t ss_tag = c->tag_seq;
    if (scan_rc == TSH_OK) {
      // (launched ahead, a group's all-gather may read a block while its rank's host still looks at it: such a
      // block is never rewritten in place -- Job::leave_overflow)
      const bool ahead = c->comm && !c->host_fn && exchange_ahead_flag().load(std::memory_order_acquire);
      scan_rc = shard_stream_begin(shard, queries + (size_t)w0 * dim, wn, k, row_mask, entries, c->d_mine, sizes[gi],
                                   /*copy_inputs=*/false, &ss, ss_tag, c->worker.get(), ahead);
      if (scan_rc) scan_err = g_err;
    }
    auto end_stream = [&] {  // nothing of this call may still run when it returns
      if (!ss) return;
      double busy = 0;
      (void)shard_stream_end(ss, &busy);
      c->scan_ns.fetch_add((int64_t)(busy * 1e3), std::memory_order_relaxed);
      ss = nullptr;
    };
    for (int32_t q0 = 0; gi < ge; q0 += sizes[gi], ++gi) {
      const int32_t gq = sizes[gi], qa = w0 + q0;  // qa: the group's first query in the call
      // the device side of the group's exchange goes out as soon as the group's scans are ENQUEUED, stream-ordered
      // behind their block writers (comm_exchange_enqueue): the collective's launch costs the host tens of
      // microseconds, which would otherwise follow the last block
      bool enqueued = false;
      if (ss && scan_rc == TSH_OK && c->comm && !c->host_fn && exchange_ahead_flag().load(std::memory_order_acquire)) {
        const double t_e = now_us();
        hipEvent_t after[MAX_CTX];
        const int n_after = shard_stream_enqueued(ss, q0 + gq, after, MAX_CTX);
        if (n_after > 0) {
          tl.wait_scan_us += now_us() - t_e;
          const double t_l = now_us();
          rc = comm_exchange_enqueue(c, c->d_mine + (size_t)q0 * bb, gq, entries, after, n_after);
          tl.pre_enqueue_us += now_us() - t_l;
          if (rc) {
            std::string keep = g_err;
            end_stream();
            g_err = keep;
            return rc;
          }
          enqueued = true;
        }
      }
      if (ss && scan_rc == TSH_OK) {
        const double t_w = now_us();
        int32_t done = 0;
        scan_rc = shard_stream_progress(ss, q0 + gq, &done);  // the group's blocks are final in d_mine
        tl.wait_scan_us += now_us() - t_w;
        if (scan_rc) scan_err = g_err;  // (the pipeline has ended: this group and the ones after it carry the error)
      }
      if (scan_rc) g_err = scan_err;
      GroupOut go;
      rc = comm_exchange_group(c, shard, c->d_mine + (size_t)q0 * bb, scan_rc, queries + (size_t)qa * dim, gq, k, thr,
                               entries, out_ids + (size_t)qa * k, out_dist + (size_t)qa * k, out_count + qa, &go, enqueued,
                               ss_tag);
      tl.groups++;
      for (int attempt = 0; rc == TSH_OK && go.need > 0; ++attempt) {
        // ties made a block overflow: every rank saw the same verdict and redoes this group with larger blocks, in a
        // buffer of their own (the pipeline keeps its block size), after the pipeline has run out: nothing overlaps a
        // retry
        if (attempt == 3) rc = set_err(TSH_E_OVERFLOW, "candidate blocks kept overflowing");
        if (rc) break;
        const double t_r = now_us();
        if (ss) {
          int32_t done = 0;
          (void)shard_stream_progress(ss, wn, &done);  // (a failure there belongs to a later group)
        }
        const int32_t ent = go.need;
        tl.retries++;
        rc = comm_reserve_agreed(c, Gmax, ent, k, 0, true, my_rows, &grew);
        if (rc) break;
        int again_rc = local_rc != TSH_OK ? local_rc : scan_rc;  // (a rank whose pipeline failed stays failed)
        std::string again_err = local_rc != TSH_OK ? local_err : scan_err;
        if (again_rc == TSH_OK) {
          const double t0 = now_us();
          again_rc = tsh_search_shard(shard, queries + (size_t)qa * dim, gq, k, row_mask, ent, c->d_retry, nullptr);
          c->scan_ns.fetch_add((int64_t)((now_us() - t0) * 1e3), std::memory_order_relaxed);
          if (again_rc) again_err = g_err;
        }
        tl.retry_scan_us += now_us() - t_r;
        if (again_rc) g_err = again_err;
        go = GroupOut();
        rc = comm_exchange_group(c, shard, c->d_retry, again_rc, queries + (size_t)qa * dim, gq, k, thr, ent,
                                 out_ids + (size_t)qa * k, out_dist + (size_t)qa * k, out_count + qa, &go);
        tl.groups++;
      }
      if (rc) {
        std::string keep = g_err;
        end_stream();
        g_err = keep;
        return rc;
      }
    }
    end_stream();
    w0 += wn;
  }
  if (trace_batch())
    fprintf(stderr, "[tsh sharded] rank %d nq=%d in %d groups (first %d, last %d): %.0f us\n", c->rank, nq, (int)sizes.size(),
            sizes.front(), sizes.back(), now_us() - t_in);
  return TSH_OK;
}

}  // extern "C"
