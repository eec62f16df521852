// CPU test of tostore_amd/csrc/tsh_host_sync.h (built and run by tests/test_host_sync.py with g++).
// Prints one "ok <name>" line per check; any failure exits non-zero with a message.
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <future>

#include "../../tostore_amd/csrc/tsh_host_sync.h"

using namespace tsh;

static void fail(const char *what) {
  fprintf(stderr, "FAIL: %s\n", what);
  exit(1);
}
template <typename F>
static bool finishes_within(F &&fn, int ms) {
  auto fut = std::async(std::launch::async, fn);
  return fut.wait_for(std::chrono::milliseconds(ms)) == std::future_status::ready;
}
static double cpu_seconds() { return (double)clock() / CLOCKS_PER_SEC; }

// The sequence ADVICE.md (round 1) describes: a ticket holds the shared lock, a writer arrives, the ticket
// holder takes the lock again before it gives the first one back.
static void test_ticket_holder_passes_waiting_writer() {
  RwLock mu;
  mu.lock_shared();  // ticket T1 (thread A)
  std::atomic<bool> writer_in{false}, writer_done{false};
  std::thread b([&] {
    mu.lock();  // append / delete (thread B): waits for T1
    writer_in = true;
    std::this_thread::sleep_for(std::chrono::milliseconds(5));
    mu.unlock();
    writer_done = true;
  });
  while (!mu.writer_pending()) std::this_thread::yield();
  if (writer_in) fail("writer entered while a reader held the lock");
  // A's next call on the same handle: with the gate it would wait for B, which waits for A
  if (!finishes_within([&] { mu.lock_shared_gate(true); }, 2000)) fail("ticket holder deadlocked behind the waiting writer");
  if (writer_in) fail("writer entered beside two readers");
  // a caller WITHOUT tickets queues behind the writer (writer preference is kept)
  std::atomic<bool> plain_in{false};
  std::thread c([&] {
    mu.lock_shared();
    plain_in = true;
    mu.unlock_shared();
  });
  std::this_thread::sleep_for(std::chrono::milliseconds(20));
  if (plain_in) fail("a reader without tickets overtook the waiting writer");
  mu.unlock_shared();  // A waits its tickets ...
  mu.unlock_shared();
  b.join();  // ... the writer runs ...
  c.join();  // ... and then the queued reader
  if (!writer_done || !plain_in) fail("writer / queued reader did not run");
  if (mu.readers() != 0 || mu.writer_pending()) fail("lock not idle at the end");
  puts("ok ticket_holder_passes_waiting_writer");
}

static void test_rwlock_exclusion_stress() {
  RwLock mu;
  std::atomic<int> inside_r{0}, inside_w{0}, bad{0};
  std::vector<std::thread> th;
  for (int t = 0; t < 6; ++t)
    th.emplace_back([&, t] {
      for (int i = 0; i < 2000; ++i) {
        if (t < 2) {
          mu.lock();
          if (inside_w.fetch_add(1) != 0 || inside_r.load() != 0) bad++;
          inside_w.fetch_sub(1);
          mu.unlock();
        } else {
          mu.lock_shared_gate((i & 7) == 0);
          inside_r.fetch_add(1);
          if (inside_w.load() != 0) bad++;
          inside_r.fetch_sub(1);
          mu.unlock_shared();
        }
      }
    });
  for (auto &x : th) x.join();
  if (bad) fail("reader and writer inside together");
  puts("ok rwlock_exclusion_stress");
}

static void test_pool_runs_every_item_once() {
  for (int n : {64, 100, 1000, 4096}) {
    std::vector<std::atomic<int>> hit((size_t)n);
    for (auto &h : hit) h = 0;
    parallel_for(n, [&](int32_t q) { hit[(size_t)q]++; });
    for (auto &h : hit)
      if (h != 1) fail("pool item not run exactly once");
  }
  puts("ok pool_runs_every_item_once");
}

// an idle pool costs nothing: no Hold, no stay_awake_until -> the workers park right after a job
static void test_pool_parks_when_idle() {
  HostPool &p = HostPool::get();
  parallel_for(256, [&](int32_t) {});
  std::this_thread::sleep_for(std::chrono::milliseconds(20));
  if (p.threads() > 1 && p.parked() != p.threads() - 1) fail("workers still polling 20 ms after a job");
  const double c0 = cpu_seconds();
  std::this_thread::sleep_for(std::chrono::milliseconds(300));
  const double burnt = cpu_seconds() - c0;
  if (burnt > 0.05) fail("idle pool burns CPU");
  {
    HostPool::Hold hold;  // the tail of a batched search: workers poll between the chunk jobs
    parallel_for(256, [&](int32_t) {});
    std::this_thread::sleep_for(std::chrono::milliseconds(5));
    if (p.threads() > 1 && p.parked() == p.threads() - 1) fail("workers parked although a Hold exists");
  }
  std::this_thread::sleep_for(std::chrono::milliseconds(20));
  if (p.threads() > 1 && p.parked() != p.threads() - 1) fail("workers still polling after the Hold ended");
  printf("ok pool_parks_when_idle (threads %d, idle cpu %.3f s)\n", p.threads(), burnt);
}

static void test_shard_workers() {
  ShardWorkers w(4);
  for (int round = 0; round < 200; ++round) {
    std::atomic<int> mask{0};
    std::thread::id ids[4];
    const std::function<void(size_t)> fn = [&](size_t g) {
      mask |= 1 << g;
      ids[g] = std::this_thread::get_id();
    };
    if (!w.run(fn)) fail("idle workers refused a job");
    if (mask != 15) fail("not every shard ran");
    if (ids[0] != std::this_thread::get_id()) fail("shard 0 must run on the caller");
  }
  // a concurrent caller is turned away (it then starts threads of its own) instead of being queued
  std::atomic<bool> inside{false}, release{false};
  const std::function<void(size_t)> slow = [&](size_t g) {
    if (g == 0) {
      inside = true;
      while (!release) std::this_thread::yield();
    }
  };
  std::thread first([&] { w.run(slow); });
  while (!inside) std::this_thread::yield();
  const std::function<void(size_t)> none = [](size_t) {};
  if (w.run(none)) fail("second caller was let in while the workers were busy");
  release = true;
  first.join();
  puts("ok shard_workers");
}

// the sharded search's look-ahead thread: jobs run one at a time, in order, wait() covers the last one
static void test_one_worker() {
  OneWorker w;
  std::vector<int> seen;
  for (int i = 0; i < 500; ++i) {
    w.post([&seen, i] { seen.push_back(i); });
    if (i % 7 == 0) w.wait();
  }
  w.wait();
  if (seen.size() != 500) fail("a job was lost");
  for (int i = 0; i < 500; ++i)
    if (seen[(size_t)i] != i) fail("jobs ran out of order");
  std::atomic<bool> ran{false};
  {
    OneWorker w2;
    w2.post([&] {
      std::this_thread::sleep_for(std::chrono::milliseconds(5));
      ran = true;
    });
  }  // the destructor lets a posted job finish
  if (!ran) fail("destructor dropped a posted job");
  puts("ok one_worker");
}

// the group schedule of a sharded call, printed for tests/test_host_sync.py to hold bench.py's mirror of it to
static void print_schedules() {
  const int nqs[] = {1, 2, 7, 8, 9, 16, 20, 37, 64, 100, 128, 129, 300, 511, 512, 1000, 5000};
  const double bytes[] = {0.0, 5.0e6, 3.0e7, 2.05e8, 3.84e8, 3.072e9, 7.68e9};
  for (int nq : nqs)
    for (double b : bytes) {
      std::vector<int32_t> sizes;
      tsh::sharded_schedule(nq, b / 6.5e6, &sizes);
      printf("schedule %d %.0f :", nq, b);
      int sum = 0, mx = 0;
      for (int32_t g : sizes) {
        printf(" %d", g);
        sum += g;
        mx = g > mx ? g : mx;
      }
      printf("\n");
      if (sum != nq || mx > tsh::sharded_group_max(nq)) {
        fprintf(stderr, "schedule of %d queries: sum %d, largest group %d > %d\n", nq, sum, mx, tsh::sharded_group_max(nq));
        exit(1);
      }
      // ... and the schedule of ranks that batch (one group up to 128 queries when the call pays for a batched pass)
      tsh::sharded_schedule(nq, b / 6.5e6, &sizes, true);
      printf("scheduleb %d %.0f :", nq, b);
      sum = mx = 0;
      for (int32_t g : sizes) {
        printf(" %d", g);
        sum += g;
        mx = g > mx ? g : mx;
      }
      printf("\n");
      if (sum != nq || mx > tsh::sharded_group_max(nq)) {
        fprintf(stderr, "batched schedule of %d queries: sum %d, largest group %d > %d\n", nq, sum, mx, tsh::sharded_group_max(nq));
        exit(1);
      }
    }
}

// a row mask as a list of row ids: the tzcnt / popcnt variant (four unconditional extractions per word) against the
// plain loop, on masks of every density; nothing is written past the count + 4 slots it is promised
static void test_mask_to_list() {
  uint64_t x = 88172645463325252ull;
  auto rnd = [&]() {
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    return x;
  };
  const int per_mille[] = {0, 1, 10, 50, 300, 700, 1000};
  const int tiles[] = {1, 2, 63, 64, 1000, 15625};
  for (int pm : per_mille)
    for (int n_tiles : tiles) {
      std::vector<uint64_t> words((size_t)n_tiles);
      for (auto &w : words) {
        w = 0;
        for (int b = 0; b < 64; ++b)
          if ((int)(rnd() % 1000) < pm) w |= 1ull << b;
      }
      if (pm == 1000) words.back() = ~0ull;  // (a full last word: five and more bits in one word)
      const int64_t bits = tsh::popcount_words(words.data(), words.size());
      int64_t naive = 0;
      for (auto w : words) naive += __builtin_popcountll(w);
      if (bits != naive) {
        fprintf(stderr, "popcount_words: %ld != %ld\n", (long)bits, (long)naive);
        exit(1);
      }
      const uint32_t CANARY = 0xC0FFEE11u;
      std::vector<uint32_t> a((size_t)bits + 4 + 16, CANARY), b((size_t)bits + 4 + 16, CANARY);
      const size_t ca = tsh::list_mask_bits_base(words.data(), n_tiles, a.data());
      size_t cb = ca;
      if (__builtin_cpu_supports("popcnt") && __builtin_cpu_supports("bmi")) cb = tsh::list_mask_bits_hw(words.data(), n_tiles, b.data());
      else b = a;
      const size_t cc = tsh::list_mask_bits(words.data(), n_tiles, bits, b.data());  // (whichever the density picks)
      bool ok = ca == (size_t)bits && cb == ca && cc == ca && std::equal(a.begin(), a.begin() + ca, b.begin());
      for (size_t i = 1; i < ca && ok; ++i) ok = a[i] > a[i - 1];
      for (size_t i = ca + 4; i < a.size() && ok; ++i) ok = a[i] == CANARY && b[i] == CANARY;
      if (!ok) {
        fprintf(stderr, "mask -> list: %d per mille, %d tiles: %zu / %zu / %zu of %ld\n", pm, n_tiles, ca, cb, cc, (long)bits);
        exit(1);
      }
    }
  printf("ok mask_to_list\n");
}

int main() {
  print_schedules();
  test_mask_to_list();
  test_ticket_holder_passes_waiting_writer();
  test_rwlock_exclusion_stress();
  test_pool_runs_every_item_once();
  test_pool_parks_when_idle();
  test_shard_workers();
  test_one_worker();
  return 0;
}
