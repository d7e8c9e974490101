// The reference's shared-memory parallel tests (test/DataStructureTest.cpp:81-120 add_remove_edge_1E5_par, :146-174
// add_remove_edge_random_2E4_par — OpenMP there, std::thread here) restated against the host shims: any number of threads
// call add_edge / remove_edge / edge_exists on one PCSR after registerThread().  Plus the same contract on a PPPCSR whose
// partitions are applied concurrently through pppcsr_apply_batch.
// Built twice: against libppcsr_hip.so (tests/test_cpp_host.py, -m gpu) and, with -fsanitize=thread, against the CPU
// emulator build of the same C ABI (tests/test_host_tsan.py) to check the shims' locking on the host side.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <thread>
#include <vector>

#include "PPPCSR.h"

static std::atomic<int> failures{0};
#define EXPECT_TRUE(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); failures++; } } while (0)
#define EXPECT_FALSE(c) EXPECT_TRUE(!(c))
#define EXPECT_EQ(a, b) do { if (!((a) == (b))) { std::printf("FAIL %s:%d: %s == %s\n", __FILE__, __LINE__, #a, #b); failures++; } } while (0)

static int scale_down = 1;  // the emulator build runs a fraction of the reference's op counts
// emulator build: narrow rounds (every emulated wave is 64 fibers; the default 6144-wide grid would take minutes)
static void tune(ppcsr_t h) {
  if (!std::getenv("PPCSR_TEST_SMALL_ROUNDS")) return;
  ppcsr_set_option(h, "mode", 0);
  ppcsr_set_option(h, "max_horizon", 32);
  ppcsr_set_option(h, "min_horizon", 4);
  ppcsr_set_option(h, "init_horizon", 8);
  ppcsr_set_option(h, "rounds_per_sync", 2);
}

template <class F>
static void parallel(int T, F f) {
  std::vector<std::thread> th;
  for (int t = 0; t < T; t++) th.emplace_back(f, t);
  for (auto &x : th) x.join();
}

static void locks_released(PCSR &pcsr) {
  for (uint32_t j = 0; j < pcsr.edges.N / pcsr.edges.logN; ++j) EXPECT_TRUE(pcsr.edges.node_locks[j]->lockable());
  EXPECT_TRUE(pcsr.edges.global_lock->lockable());
}

static void run(bool lock_search, int T) {
  {  // add_remove_edge_1E5_par: every thread inserts its share of (0, i), checking a sample right after the insert
    PCSR pcsr(10, 10, lock_search, 0);
    tune(pcsr.handle());
    const int edge_count = 100000 / scale_down;
    const int sample = 211;  // every edge_exists is a flush + a device round trip
    parallel(T, [&](int t) {
      pcsr.edges.global_lock->registerThread();
      for (int i = 1 + t; i < edge_count + 1; i += T) {
        pcsr.add_edge(0, i, i);
        if (i % sample == 0) EXPECT_TRUE(pcsr.edge_exists(0, i));
      }
      pcsr.edges.global_lock->unregisterThread();
    });
    locks_released(pcsr);
    EXPECT_EQ(pcsr.get_n(), 10u);
    EXPECT_EQ(pcsr.getNode(0).num_neighbors, (uint32_t)edge_count);
    EXPECT_EQ(pcsr.get_neighbourhood(0).size(), (size_t)edge_count);
    parallel(T, [&](int t) {
      pcsr.edges.global_lock->registerThread();
      for (int i = 1 + t; i < edge_count + 1; i += T) {
        pcsr.remove_edge(0, i);
        if (i % sample == 0) EXPECT_FALSE(pcsr.edge_exists(0, i));
      }
      pcsr.edges.global_lock->unregisterThread();
    });
    locks_released(pcsr);
    EXPECT_EQ(pcsr.get_neighbourhood(0).size(), 0u);
    EXPECT_EQ(pcsr.get_n(), 10u);
    EXPECT_EQ(pcsr.edges.global_lock->registered.load(), 0);
  }
  {  // add_remove_edge_random_2E4_par (2e5 ops): 75 % add / 25 % delete of random pairs.  Thread t only touches targets
     // congruent to t mod T, so the check right after each op is exact (the reference's shared std::rand() makes its own
     // assertions racy: another thread may delete the pair in between)
    PCSR pcsr(1000, 1000, lock_search, 0);
    tune(pcsr.handle());
    const int edge_count = 200000 / scale_down;
    std::vector<std::set<std::pair<int, int>>> live(T);
    parallel(T, [&](int t) {
      pcsr.edges.global_lock->registerThread();
      uint64_t x = 88172645463325252ull + 7919ull * (uint64_t)t;
      auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 11); };
      for (int i = 1 + t; i < edge_count + 1; i += T) {
        const int src = rnd() % 1000, target = (int)((rnd() % (1000 / T)) * T + t) % 1000;
        if (rnd() % 4 != 0) {
          pcsr.add_edge(src, target, i);
          live[t].insert({src, target});
          if (i % 173 == 0) EXPECT_TRUE(pcsr.edge_exists(src, target));
        } else {
          pcsr.remove_edge(src, target);
          live[t].erase({src, target});
          if (i % 173 == 0) EXPECT_FALSE(pcsr.edge_exists(src, target));
        }
      }
      pcsr.edges.global_lock->unregisterThread();
    });
    locks_released(pcsr);
    size_t expect = 0, have = 0;
    for (auto &s : live) expect += s.size();
    for (int v = 0; v < 1000; v++) have += pcsr.get_neighbourhood(v).size();
    EXPECT_EQ(have, expect);
    for (int t = 0; t < T; t++) {
      int k = 0;
      for (auto &e : live[t])
        if (k++ % 97 == 0) EXPECT_TRUE(pcsr.edge_exists(e.first, e.second));
    }
  }
  {  // the same contract on a PPPCSR (4 partitions on one GPU): writers register with the partition they write to
     // (thread_pool_pppcsr.cpp:60-66); the batch is bucketed by owner and the partitions are applied concurrently
    PPPCSR pp(1000, 1000, lock_search, 1, 4, false);
    for (int p = 0; p < 4; p++) tune(pp.partition(p).handle());
    const int edge_count = 60000 / scale_down;
    std::vector<std::set<std::pair<int, int>>> live(T);
    parallel(T, [&](int t) {
      for (int p = 0; p < 4; p++) pp.registerThread(p);
      uint64_t x = 1234567ull + 104729ull * (uint64_t)t;
      auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 11); };
      for (int i = 1 + t; i < edge_count + 1; i += T) {
        const int src = rnd() % 1000, target = (int)((rnd() % (1000 / T)) * T + t) % 1000;
        if (rnd() % 5 != 0) {
          pp.add_edge(src, target, 1);
          live[t].insert({src, target});
        } else {
          pp.remove_edge(src, target);
          live[t].erase({src, target});
        }
        if (i % 499 == 0) EXPECT_EQ(pp.edge_exists(src, target), live[t].count({src, target}) != 0);
      }
      for (int p = 0; p < 4; p++) pp.unregisterThread(p);
    });
    size_t expect = 0, have = 0;
    for (auto &s : live) expect += s.size();
    for (int v = 0; v < 1000; v++) have += pp.get_neighbourhood(v).size();
    EXPECT_EQ(have, expect);
    EXPECT_EQ(pp.get_n(), 1000u);
  }
}

int main(int argc, char **argv) {
  PCSR::quiet() = true;
  int T = 8;
  if (argc > 1) T = std::atoi(argv[1]);
  if (argc > 2) scale_down = std::atoi(argv[2]);
  if (T < 1 || 1000 % T != 0) T = 8;
  run(true, T);
  run(false, T);
  std::printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures.load());
  return failures ? 1 : 0;
}
