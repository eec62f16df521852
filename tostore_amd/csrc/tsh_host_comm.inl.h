// tsh_host_comm.inl.h -- the RCCL exchange of a row-sharded index, reachable from the C ABI (one process per GPU)
// Part of the single translation unit tsh_lib.hip (textually included there; not compiled alone).
//
// BASELINE.json's north star: "the corpus shards by row-range across the 8 GPUs of one node with a RCCL
// all-gather of per-shard (distance, row-id) top-k candidates over xGMI and a final host-side merge".  The Python
// harness does that exchange with torch.distributed (tostore_amd/sharded.py); a Dart host has no torch, so the
// same three steps -- shard scan into device blocks, ncclAllGather of the blocks, host merge -- are offered here
// behind plain C entry points.  librccl is loaded with dlopen on first use: a process that never shards does not
// pull it in, and a process that already holds a librccl (a torch host) gets that one.
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
};

RcclApi *rccl() {
  static RcclApi *api = [] {
    RcclApi *a = new RcclApi();
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
    }
    if (!h) {
      a->err = std::string("librccl not found: ") + (dlerror() ? dlerror() : "");
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

}  // namespace

struct tsh_comm {
  void *comm = nullptr;  // ncclComm_t
  int32_t world = 1, rank = 0, device = 0;
  hipStream_t stream = nullptr;
  std::mutex mu;  // collectives of one communicator are issued one call at a time
  uint8_t *d_mine = nullptr, *d_all = nullptr, *h_all = nullptr;
  size_t mine_cap = 0, all_cap = 0;
};

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
  if (!out) return set_err(TSH_E_BAD_ARG, "out is NULL");
  *out = nullptr;
  if (!id || world < 1 || rank < 0 || rank >= world) return set_err(TSH_E_BAD_ARG, "bad id / world / rank");
  RcclApi *r = rccl();
  if (!r->ok) return set_err(TSH_E_RCCL, "%s", r->err.c_str());
  if (device_count_cached() <= 0) return set_err(TSH_E_NO_DEVICE, "no HIP device available");
  if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
  if (device >= device_count_cached()) return set_err(TSH_E_BAD_ARG, "device %d not present", device);
  HIPCHK(hipSetDevice(device));
  std::unique_ptr<tsh_comm> c(new tsh_comm());
  c->world = world;
  c->rank = rank;
  c->device = device;
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  RcclId uid;
  memcpy(&uid, id, sizeof uid);
  int rc = r->CommInitRank(&c->comm, world, uid, rank);  // collective: every rank of the job calls it
  if (rc != 0) {
    hipStreamDestroy(c->stream);
    return rccl_fail("ncclCommInitRank", rc);
  }
  *out = c.release();
  return TSH_OK;
}

int32_t tsh_comm_destroy(tsh_comm *c) {
  if (!c) return TSH_OK;
  hipSetDevice(c->device);
  if (c->comm) rccl()->CommDestroy(c->comm);
  hipFree(c->d_mine);
  hipFree(c->d_all);
  hipHostFree(c->h_all);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
  return TSH_OK;
}

int32_t tsh_comm_world(tsh_comm *c) { return c ? c->world : 0; }

int32_t tsh_search_sharded(tsh_index *shard, tsh_comm *c, const float *queries, int32_t nq, int32_t k, double thr,
                           const uint8_t *row_mask, int64_t *out_ids, double *out_dist, int32_t *out_count) {
  if (!shard || !c) return set_err(TSH_E_BAD_ARG, "shard / comm is NULL");
  if (shard->shards.size() != 1) return set_err(TSH_E_BAD_ARG, "needs a single-shard handle (tsh_index_create_shard)");
  if (nq < 0) return set_err(TSH_E_BAD_ARG, "nq < 0");
  if (nq == 0) return TSH_OK;
  if (!queries || !out_count) return set_err(TSH_E_BAD_ARG, "queries / out_count is NULL");
  for (int32_t q = 0; q < nq; ++q) out_count[q] = 0;
  if (k <= 0) return TSH_OK;
  if (!out_ids || !out_dist) return set_err(TSH_E_BAD_ARG, "out_ids / out_dist is NULL");
  if (shard->shards[0]->device != c->device) return set_err(TSH_E_BAD_ARG, "shard and communicator sit on different devices");
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(hipSetDevice(c->device));
  int32_t entries = tsh_default_block_entries(k);
  for (int attempt = 0; attempt < 3; ++attempt) {
    const size_t bb = (size_t)tsh_candidate_block_bytes(entries), mine = bb * (size_t)nq, all = mine * (size_t)c->world;
    if (mine > c->mine_cap) {
      hipFree(c->d_mine);
      c->d_mine = nullptr;
      c->mine_cap = 0;
      HIPCHK(hipMalloc(&c->d_mine, mine));
      c->mine_cap = mine;
    }
    if (all > c->all_cap) {
      hipFree(c->d_all);
      hipHostFree(c->h_all);
      c->d_all = c->h_all = nullptr;
      c->all_cap = 0;
      HIPCHK(hipMalloc(&c->d_all, all));
      HIPCHK(hipHostMalloc(&c->h_all, all, hipHostMallocDefault));
      c->all_cap = all;
    }
    // 1. this rank's shard: candidate blocks stay in device memory (host-synchronised on return)
    int rc = tsh_search_shard(shard, queries, nq, k, row_mask, entries, c->d_mine, nullptr);
    if (rc != TSH_OK) return rc;
    // 2. all-gather over RCCL (xGMI between the GPUs of a node): k' x 24 B per rank and query -- latency-bound
    int nrc = rccl()->AllGather(c->d_mine, c->d_all, mine, /*ncclChar*/ 0, c->comm, c->stream);
    if (nrc != 0) return rccl_fail("ncclAllGather", nrc);
    HIPCHK(hipMemcpyAsync(c->h_all, c->d_all, all, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    // 3. host merge; a truncated block (ties) asks every rank alike for a retry with more entries
    int32_t need = entries;
    rc = tsh_merge_candidates(shard->metric, shard->dim, queries, nq, k, thr, c->h_all, c->world, entries, out_ids, out_dist,
                              out_count, &need);
    if (rc != TSH_E_OVERFLOW) return rc;
    entries = need;
  }
  return set_err(TSH_E_OVERFLOW, "candidate blocks kept overflowing");
}

}  // extern "C"
