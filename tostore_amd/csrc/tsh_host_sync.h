// tsh_host_sync.h -- host-side synchronisation pieces of libtostore_hip.so: the handle lock, the pool that
// finalises batches, the per-shard workers of an in-process multi-GPU handle.  Plain C++17 (no HIP) so that
// tests/test_host_sync.py can compile and exercise it with g++ on a machine without a GPU.
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace tsh {

inline double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// one turn of a polling loop: the x86 pause hint where there is one, a yield elsewhere (an aarch64 ROCm host builds
// this header too)
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#else
  std::this_thread::yield();
#endif
}

// Reader/writer lock that may be released by a different thread than the one that took it (an asynchronous
// search is submitted and waited independently).  Writers have preference: a reader that arrives while a writer
// waits queues behind it -- EXCEPT a reader that says its handle already holds shared locks which it can only
// give back after this call (open asynchronous tickets).  That reader would wait for the writer, the writer for
// the ticket, and the ticket for the reader: it passes the gate instead (the writer cannot run before the ticket
// is waited anyway).  So that a caller who always has a ticket open cannot starve the writer, tsh_search_submit
// refuses new tickets (TSH_E_BUSY) while writer_pending() and tickets are open: the pipeline drains, the
// writer runs.
class RwLock {
 public:
  void lock_shared() { lock_shared_gate(false); }
  void lock_shared_gate(bool bypass_gate) {
    std::unique_lock<std::mutex> lk(m_);
    cv_.wait(lk, [&] { return !writer_ && (bypass_gate || writers_waiting_ == 0); });
    ++readers_;
  }
  void unlock_shared() {
    std::lock_guard<std::mutex> lk(m_);
    if (--readers_ == 0) cv_.notify_all();
  }
  void lock() {
    std::unique_lock<std::mutex> lk(m_);
    ++writers_waiting_;
    cv_.wait(lk, [&] { return !writer_ && readers_ == 0; });
    --writers_waiting_;
    writer_ = true;
  }
  void unlock() {
    std::lock_guard<std::mutex> lk(m_);
    writer_ = false;
    cv_.notify_all();
  }
  bool writer_pending() {
    std::lock_guard<std::mutex> lk(m_);
    return writer_ || writers_waiting_ > 0;
  }
  int readers() {
    std::lock_guard<std::mutex> lk(m_);
    return readers_;
  }

 private:
  std::mutex m_;
  std::condition_variable cv_;
  int readers_ = 0, writers_waiting_ = 0;
  bool writer_ = false;
};

// CPUs the container may use per scheduling period (cgroup v2 cpu.max, v1 cpu.cfs_quota_us); 0 = no limit known.
// It can be far below the core count the box shows (16 of 256 on the MI355X boxes measured here), and a process
// group that runs past it is frozen for the rest of the 100 ms period.
inline unsigned cpu_quota_cpus() {
  unsigned quota = 0;
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
    long long q = 0, period = 0;
    if (fscanf(f, "%lld %lld", &q, &period) == 2 && q > 0 && period > 0) quota = (unsigned)std::max<long long>(1, q / period);
    fclose(f);
  } else if (FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1 (-1 = unlimited)
    long long q = 0, period = 0;
    if (fscanf(fq, "%lld", &q) != 1) q = 0;
    fclose(fq);
    if (FILE *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
      if (fscanf(fp, "%lld", &period) != 1) period = 0;
      fclose(fp);
    }
    if (q > 0 && period > 0) quota = (unsigned)std::max<long long>(1, q / period);
  }
  return quota;
}
// ---- what the environment may and may not steer -------------------------------------------------------------
// The library lives inside a database process: its environment is not ours to obey.  Release builds read four
// variables, all about how the HOST side waits or logs (LOCAL_WORLD_SIZE, TSH_BLOCKING_WAIT, TSH_HOST_THREADS,
// TSH_TRACE_BATCH).  Everything else is one of two kinds:
//   probe_env(name)  experiment switches (kernel shapes, stream layouts, sample sizes ...): exist in probe builds
//                    only (-DTSH_PROBES, tools/build_variants.py); a release build answers "not set";
//   test_env(name)   test hooks that change what the library loads or make it fail on purpose (TSH_RCCL_LIB,
//                    TSH_TEST_FAIL_ALLOC_OVER, TSH_SHARDS_SHARE_DEVICES): read only after the PROCESS itself asked
//                    for them, tsh_index_set_option(NULL, TSH_OPT_TEST_HOOKS, TSH_TEST_HOOKS_MAGIC) -- the variables
//                    alone change nothing (tests/test_abi.py).
inline const char *probe_env(const char *name) {
#ifdef TSH_PROBES
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}
inline std::atomic<bool> &test_hooks_flag() {
  static std::atomic<bool> on{false};
  return on;
}
inline std::atomic<bool> &exchange_ahead_flag() {  // TSH_OPT_EXCHANGE_AHEAD (off: measured slower)
  static std::atomic<bool> on{false};
  return on;
}
inline const char *test_env(const char *name) { return test_hooks_flag().load(std::memory_order_acquire) ? getenv(name) : nullptr; }

inline unsigned local_peers() {  // processes of this job on this node (one per GPU): torchrun / bench.py export it
  const char *p = getenv("LOCAL_WORLD_SIZE");
  return p && atoi(p) > 0 ? (unsigned)atoi(p) : 1u;
}
// How a thread waits for the GPU.  The HIP runtime's default is to spin: fine for one process on a box of its
// own, but a rank per GPU spins in two or three threads (the caller, the sharded search's look-ahead, a submit
// thread), and eight ranks then burn the whole CPU quota of a 16-CPU container while waiting -- the throttling
// that follows freezes every rank for tens of milliseconds.  With fewer than three CPUs of quota per rank the
// library's completion events block (interrupt) instead; TSH_BLOCKING_WAIT=0 / 1 overrides.
inline bool blocking_wait() {
  static const bool v = [] {
    if (const char *e = getenv("TSH_BLOCKING_WAIT")) return e[0] == '1';
    const unsigned q = cpu_quota_cpus(), peers = local_peers();
    return peers > 1 && q > 0 && q < 3 * peers;
  }();
  return v;
}

// run fn(q) for q in [0,n) on a few host threads (per-query preparation / finalisation of a batch).  The
// workers are created once and parked on a condition variable: spawning threads per call costs more than the
// work itself (about 30 us per thread on a 128-core host).
//
// Polling policy.  The library lives inside the database's own process, so an idle pool must cost nothing: a
// worker parks the moment it finds no work -- unless a library call that is about to hand over more jobs has
// said so: HostPool::Hold (the tail of a batched search: one job per chunk of queries, a few hundred
// microseconds apart, and waking a parked thread costs more than such a job) or stay_awake_until (a GPU wait
// the caller expects to be shorter than a millisecond).  Both end with the call.  TSH_HOST_SPIN_US=n adds n
// microseconds of polling after every job (default 0; measurements only).
class HostPool {
 public:
  static HostPool &get() {
    static HostPool *p = new HostPool();  // never destroyed: workers may outlive static teardown
    return *p;
  }
  // false when the pool is busy with another caller's job (the caller then runs inline).
  // A job is complete when all its ITEMS are done, not when every worker has reported: a worker that wakes up
  // late (a parked thread needs 30-50 us, a descheduled one milliseconds) finds the job closed and goes back to
  // waiting -- the caller and the workers that are awake have done its share.
  bool run(int32_t n, const std::function<void(int32_t)> &fn) {
    std::unique_lock<std::mutex> own(owner_, std::try_to_lock);
    if (!own.owns_lock()) return false;
    fn_ = &fn;
    n_ = n;
    grain_ = std::max(1, std::min(8, n / (4 * threads())));  // a 128-item job on 32 threads: items of one, not of eight
    done_items_.store(0, std::memory_order_relaxed);
    next_.store(0, std::memory_order_relaxed);
    open_.store(true, std::memory_order_seq_cst);
    {
      std::lock_guard<std::mutex> lk(m_);  // a worker between its last look at gen_ and cv_.wait must not miss this
      gen_.fetch_add(1, std::memory_order_release);
    }
    if (parked_.load(std::memory_order_acquire) > 0) cv_.notify_all();
    chunks();
    spin_until([&] { return done_items_.load(std::memory_order_acquire) >= n; });
    // store-then-load on this side, add-then-load on the worker's (loop()): all four sequentially consistent, or
    // this thread could read active_ == 0 while a late worker still reads open_ == true and walks into the next job
    open_.store(false, std::memory_order_seq_cst);
    spin_until([&] { return active_.load(std::memory_order_seq_cst) == 0; });  // nobody still looks at fn_ / n_
    fn_ = nullptr;
    return true;
  }
  int threads() const { return (int)workers_.size() + 1; }
  int parked() const { return parked_.load(std::memory_order_acquire); }
  // A caller about to wait for the GPU and then hand the pool a job: keep the workers polling until `t_us`
  // (now_us() clock) so the job does not start with waking them.  Bounded by the caller (a millisecond at most).
  void stay_awake_until(double t_us) {
    double cur = awake_until_.load(std::memory_order_relaxed);
    while (t_us > cur && !awake_until_.compare_exchange_weak(cur, t_us, std::memory_order_relaxed)) {
    }
  }
  // While a Hold exists the workers poll between jobs instead of parking.  Scope it to the stretch of ONE library
  // call in which jobs follow each other closely.
  class Hold {
   public:
    Hold() {
      HostPool &p = get();
      p.holders_.fetch_add(1, std::memory_order_acq_rel);
      if (p.parked_.load(std::memory_order_acquire) > 0) {  // wake the parked workers: they find no open job and poll
        {
          std::lock_guard<std::mutex> lk(p.m_);
          p.gen_.fetch_add(1, std::memory_order_release);
        }
        p.cv_.notify_all();
      }
    }
    ~Hold() { get().holders_.fetch_sub(1, std::memory_order_acq_rel); }
    Hold(const Hold &) = delete;
    Hold &operator=(const Hold &) = delete;
  };

 private:
  HostPool() {
    // one process per GPU shares the cores with its peers (torchrun exports LOCAL_WORLD_SIZE); TSH_HOST_THREADS
    // overrides
    unsigned hw = std::thread::hardware_concurrency();
    // A container's CPU quota (cgroup v2 cpu.max) can be far below the core count it shows (16 of 256 on the
    // MI355X boxes measured here), and a process that runs past it is frozen for the rest of the 100 ms period
    // -- seen as one 60 ms stall every few hundred calls with 63 polling workers.  Workers only poll while a
    // call holds them (Hold, ~20 % of a 1024-query call), so twice the quota is the limit that stayed clear of it.
    const unsigned quota = cpu_quota_cpus();
    if (quota) hw = std::min(hw, 2 * quota);
    hw = std::max(1u, hw / local_peers());
    // (a 256-query chunk of a batch's tail is ~1.5 ms of single-thread finalisation: 32 threads take 45-55 us
    // over it, as fast as the GPU delivers chunks)
    int nt = (int)std::min<unsigned>(hw > 1 ? hw - 1 : 0, hw >= 32 ? 31 : 15);
    if (const char *forced = getenv("TSH_HOST_THREADS")) nt = std::max(0, std::min(atoi(forced) - 1, 63));
    if (const char *spin = probe_env("TSH_HOST_SPIN_US")) spin_us_ = std::max(0.0, atof(spin));
    for (int i = 0; i < nt; ++i) {
      workers_.emplace_back([this] { loop(); });
      workers_.back().detach();
    }
  }
  template <typename F>
  static void spin_until(F &&cond) {
    for (int spins = 0; !cond(); ++spins) {
      if (spins < 20000) cpu_relax();
      else std::this_thread::yield();
    }
  }
  void chunks() {
    for (;;) {
      const int32_t g = grain_;
      int32_t q0 = next_.fetch_add(g, std::memory_order_acq_rel);
      if (q0 >= n_) return;
      const int32_t q1 = std::min(n_, q0 + g);
      for (int32_t q = q0; q < q1; ++q) (*fn_)(q);
      done_items_.fetch_add(q1 - q0, std::memory_order_acq_rel);
    }
  }
  bool keep_polling(double t0) const {
    if (holders_.load(std::memory_order_acquire) > 0) return true;
    const double t = now_us();
    return t < awake_until_.load(std::memory_order_relaxed) || t - t0 < spin_us_;
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      const double t0 = now_us();
      bool got = false;
      for (int i = 0;; ++i) {
        if (gen_.load(std::memory_order_acquire) != seen) {
          got = true;
          break;
        }
        if ((i & 63) == 0 && !keep_polling(t0)) break;
        cpu_relax();
      }
      if (!got) {
        std::unique_lock<std::mutex> lk(m_);
        parked_.fetch_add(1, std::memory_order_acq_rel);
        cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
        parked_.fetch_sub(1, std::memory_order_acq_rel);
      }
      seen = gen_.load(std::memory_order_acquire);
      active_.fetch_add(1, std::memory_order_seq_cst);
      if (open_.load(std::memory_order_seq_cst)) chunks();  // closed: the job finished without this worker
      active_.fetch_sub(1, std::memory_order_acq_rel);
    }
  }
  std::vector<std::thread> workers_;
  std::mutex owner_, m_;
  std::condition_variable cv_;
  const std::function<void(int32_t)> *fn_ = nullptr;
  int32_t n_ = 0, grain_ = 8;
  double spin_us_ = 0.0;
  std::atomic<int32_t> next_{0}, done_items_{0};
  std::atomic<int> active_{0}, parked_{0}, holders_{0};
  std::atomic<bool> open_{false};
  std::atomic<double> awake_until_{0.0};
  std::atomic<uint64_t> gen_{0};
};

template <typename F>
void parallel_for(int32_t n, F fn) {
  if (n >= 24) {  // (below that the calling thread is done before the workers are up)
    std::function<void(int32_t)> f = fn;
    if (HostPool::get().run(n, f)) return;
  }
  for (int32_t q = 0; q < n; ++q) fn(q);
}
template <typename F>
void parallel_for_range(int32_t q0, int32_t q1, F fn) {
  parallel_for(q1 - q0, [&](int32_t i) { fn(q0 + i); });
}

// The shards of an in-process multi-GPU handle (tsh_index_create with n_devices > 1) are searched side by
// side, one host thread per shard.  The threads live as long as the handle (starting one costs ~30 us, a
// 125 k-row shard scan 70 us): run() hands shard g's share of a call to worker g and does shard 0 itself.
class ShardWorkers {
 public:
  explicit ShardWorkers(int n_shards) {
    for (int g = 1; g < n_shards; ++g) slots_.emplace_back(new Slot());
    for (auto &s : slots_) s->th = std::thread([sp = s.get()] { sp->loop(); });
  }
  ~ShardWorkers() {
    for (auto &s : slots_) {
      {
        std::lock_guard<std::mutex> lk(s->m);
        s->stop = true;
      }
      s->cv.notify_all();
      if (s->th.joinable()) s->th.join();
    }
  }
  // fn(g) for g in [0, n_shards): g = 0 on the calling thread, the others on their workers.  One call at a time
  // uses the workers; a concurrent caller (try_lock fails) gets false and starts threads of its own.
  bool run(const std::function<void(size_t)> &fn) {
    std::unique_lock<std::mutex> own(owner_, std::try_to_lock);
    if (!own.owns_lock()) return false;
    for (size_t i = 0; i < slots_.size(); ++i) {
      Slot &s = *slots_[i];
      {
        std::lock_guard<std::mutex> lk(s.m);
        s.fn = &fn;
        s.g = i + 1;
        s.pending = true;
      }
      s.cv.notify_all();
    }
    fn(0);
    for (auto &sp : slots_) {
      std::unique_lock<std::mutex> lk(sp->m);
      sp->cv.wait(lk, [&] { return !sp->pending; });
    }
    return true;
  }

 private:
  struct Slot {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    const std::function<void(size_t)> *fn = nullptr;
    size_t g = 0;
    bool pending = false, stop = false;
    void loop() {
      std::unique_lock<std::mutex> lk(m);
      for (;;) {
        cv.wait(lk, [&] { return pending || stop; });
        if (stop) return;
        const std::function<void(size_t)> *f = fn;
        const size_t gg = g;
        lk.unlock();
        (*f)(gg);
        lk.lock();
        pending = false;
        cv.notify_all();
      }
    }
  };
  std::mutex owner_;
  std::vector<std::unique_ptr<Slot>> slots_;
};

// ---- a row mask as a list of row ids (tsh_lib.hip build_row_list) ----------------------------------------------
// The set bits of a mask as ascending positions -> out (room for their count + 4), their number.  A lone masked query
// waits for this on the host: with push_back and one loop exit per word (mispredicted every other word at 1 %) a
// 1 M-row mask took 139 us on a 2.1 GHz core, its popcount without the instruction 39 -- now 43 and 10
// (four unconditional extractions per word, tzcnt(0) = 64 writing a slot the next word overwrites).
#if defined(__x86_64__)
__attribute__((target("popcnt,bmi"))) inline size_t list_mask_bits_hw(const uint64_t *words, int32_t n_tiles, uint32_t *out) {
  uint32_t *o = out;
  for (int32_t t = 0; t < n_tiles; ++t) {
    uint64_t w = words[(size_t)t];
    const uint32_t base = (uint32_t)t * 64u;
    const int c = __builtin_popcountll(w);
    o[0] = base + (uint32_t)__builtin_ia32_tzcnt_u64(w);
    w &= w - 1;
    o[1] = base + (uint32_t)__builtin_ia32_tzcnt_u64(w);
    w &= w - 1;
    o[2] = base + (uint32_t)__builtin_ia32_tzcnt_u64(w);
    w &= w - 1;
    o[3] = base + (uint32_t)__builtin_ia32_tzcnt_u64(w);
    w &= w - 1;
    if (c > 4) {
      uint32_t *p = o + 4;
      for (; w; w &= w - 1) *p++ = base + (uint32_t)__builtin_ctzll(w);
    }
    o += c;
  }
  return (size_t)(o - out);
}
#endif
inline size_t list_mask_bits_base(const uint64_t *words, int32_t n_tiles, uint32_t *out) {
  uint32_t *o = out;
  for (int32_t t = 0; t < n_tiles; ++t)
    for (uint64_t w = words[(size_t)t]; w; w &= w - 1) *o++ = (uint32_t)t * 64u + (uint32_t)__builtin_ctzll(w);
  return (size_t)(o - out);
}
inline size_t list_mask_bits(const uint64_t *words, int32_t n_tiles, int64_t bits, uint32_t *out) {
#if defined(__x86_64__)
  static const bool hw = __builtin_cpu_supports("popcnt") && __builtin_cpu_supports("bmi");
  // (hardly any word has a bit: the plain loop's exits are predictable and it does nothing per empty word)
  if (hw && bits * 8 > n_tiles) return list_mask_bits_hw(words, n_tiles, out);
#else
  (void)bits;
#endif
  return list_mask_bits_base(words, n_tiles, out);
}
#if defined(__x86_64__)
__attribute__((target("popcnt"))) inline int64_t popcount_words_hw(const uint64_t *w, size_t n) {
  int64_t r = 0;
  for (size_t i = 0; i < n; ++i) r += __builtin_popcountll(w[i]);
  return r;
}
#endif
inline int64_t popcount_words(const uint64_t *w, size_t n) {
#if defined(__x86_64__)
  static const bool hw = __builtin_cpu_supports("popcnt");
  if (hw) return popcount_words_hw(w, n);
#endif
  int64_t r = 0;
  for (size_t i = 0; i < n; ++i) r += __builtin_popcountll(w[i]);
  return r;
}

// ---- how tsh_search_sharded cuts a call into groups (tsh_host_comm.inl.h) --------------------------------------
constexpr int32_t SHARDED_WINDOW = 4096;  // queries of a call whose blocks this rank keeps at once
// largest group any schedule of an nq-query call can contain (what the buffers are sized for: capacities must not
// depend on anything a rank knows alone)
inline int32_t sharded_group_max(int32_t nq) {
  // (a group of a big call is one batched call on every shard, with its own sample pass, selects and re-rank: on a
  // 1.25 M x 1536 shard 256 queries cost 1.2 ms where 1024 in one call cost 4.0, on 1 M x 768 five groups of 64 cost
  // 2.6 ms where 300 queries in groups of 256 cost 1.3 -- tools/r6_sharded_batch_probe.py)
  if (nq >= 1024) return 512;
  if (nq > 128) return 256;
  return std::max(nq, 1);
}
// Does a call of nq queries go to the matrix cores on shards whose single-query scan takes scan_us?  The schedule's
// copy of shard_takes_batch's cost model (tsh_host_batch.inl.h: 0.30 ms + a pass over the fp16 copy at ~4 TB/s per 128
// queries, against a pipelined scan at ~6.6 TB/s + 25 us per query), in terms of what every rank knows.
inline bool sharded_call_batches(int32_t nq, double scan_us) {
  if (nq < 2) return false;
  const double t_batch = 300.0 + 0.825 * scan_us * (double)((nq + 127) / 128);
  const double t_single = (double)nq * (scan_us + 25.0);
  return t_single > t_batch;
}
// Queries per exchange, group after group.
// Scanned one by one (batched == false: no rank batches, or the cost model says scans), the scans of a call run as ONE
// pipeline whatever the groups are, so a group costs its exchange only (collective latency + copy + merge: ~0.1 ms):
// hidden behind the scans of the groups after it, exposed for the LAST group.  Up to 128 queries the groups therefore
// shrink -- half of what is left each time, never below what it takes to hide an exchange (scan_us: one query's scan on
// the largest shard, from the bytes the ranks told each other at their last agreement) nor below four, the rest in one
// piece once it is that small: 20 queries on 125 k x 768 shards go as 10 + 5 + 5, on 10 k-row shards (a scan is shorter
// than any exchange) as one group.
// On the matrix cores (batched: the ranks batch calls, and this one pays for it) a group is one batched call per shard
// -- its own sample pass, selects and re-rank, ~0.3 ms before the first row is scored: a call of up to 128 queries is
// ONE group (three groups of 10 + 5 + 5 would be three batched calls), bigger ones go in big uniform groups.
inline void sharded_schedule(int32_t nq, double scan_us, std::vector<int32_t> *sizes, bool batched = false) {
  sizes->clear();
  if (nq > 128) {
    const int32_t G = sharded_group_max(nq);
    for (int32_t q = 0; q < nq; q += G) sizes->push_back(std::min(G, nq - q));
    return;
  }
  if (batched && sharded_call_batches(nq, scan_us)) {
    sizes->push_back(nq);
    return;
  }
  const double exchange_us = 150.0;
  const int32_t g_min = (int32_t)std::min(128.0, std::max(4.0, std::ceil(exchange_us / std::max(scan_us, 1.0))));
  int32_t rem = nq;
  while (rem > 0) {
    const int32_t g = rem < 2 * g_min ? rem : std::max(g_min, (rem + 1) / 2);
    sizes->push_back(g);
    rem -= g;
  }
}

// One persistent helper thread that runs one job at a time (the sharded search's look-ahead: group g + 1 is scanned
// while the caller exchanges and merges group g).  post() hands it a job, wait() returns when that job is done.
class OneWorker {
 public:
  // spin_us: after a job the thread polls for the next one that long before it parks (a caller that comes back within
  // microseconds -- the regions of a benchmark, a host streaming queries -- then does not pay a futex wake-up, 20-50 us,
  // in front of its first scan); 0 = park at once
  explicit OneWorker(double spin_us = 0) : spin_us_(spin_us), th_([this] { loop(); }) {}
  ~OneWorker() {
    stopping_.store(true, std::memory_order_release);  // (the polling loop looks at this one: it holds no lock)
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    if (th_.joinable()) th_.join();
  }
  void post(std::function<void()> fn) {
    std::unique_lock<std::mutex> lk(m_);
    cv_.wait(lk, [&] { return !pending_; });
    fn_ = std::move(fn);
    pending_ = true;
    posted_.store(true, std::memory_order_release);
    lk.unlock();
    cv_.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(m_);
    cv_.wait(lk, [&] { return !pending_; });
  }

 private:
  void loop() {
    std::unique_lock<std::mutex> lk(m_);
    for (;;) {
      if (spin_us_ > 0 && !pending_ && !stop_) {
        lk.unlock();
        const double t_end = now_us() + spin_us_;
        while (!posted_.load(std::memory_order_acquire) && !stopping_.load(std::memory_order_acquire) && now_us() < t_end)
          cpu_relax();
        lk.lock();
      }
      cv_.wait(lk, [&] { return pending_ || stop_; });
      if (!pending_) return;  // (a job posted before the destructor still runs)
      posted_.store(false, std::memory_order_relaxed);
      std::function<void()> f = std::move(fn_);
      lk.unlock();
      f();
      lk.lock();
      pending_ = false;
      cv_.notify_all();
    }
  }
  std::mutex m_;
  std::condition_variable cv_;
  std::function<void()> fn_;
  bool pending_ = false, stop_ = false;
  std::atomic<bool> posted_{false}, stopping_{false};
  double spin_us_ = 0;
  std::thread th_;  // last member: the thread starts with everything above constructed
};

}  // namespace tsh
