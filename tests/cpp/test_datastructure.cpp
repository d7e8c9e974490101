// The reference's gtest suite (test/DataStructureTest.cpp:12-213) restated on a minimal assert harness against the
// host shims (gtest is not installed).  Same workloads, same assertions; the lock-release checks go through the
// inert lock stand-ins so the test source reads like the reference's.  Built and run by tests/test_cpp_host.py (-m gpu).
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "PPPCSR.h"
#include "bfs.h"
#include "pagerank.h"

static int failures = 0;
#define EXPECT_TRUE(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); failures++; } } while (0)
#define EXPECT_FALSE(c) EXPECT_TRUE(!(c))
#define EXPECT_EQ(a, b) do { if (!((a) == (b))) { std::printf("FAIL %s:%d: %s == %s\n", __FILE__, __LINE__, #a, #b); failures++; } } while (0)

// the reference's consumers: engine-backed PCSR -> GPU (host/bfs.h, host/pagerank.h); the templates below are the host
// statements of src/utility/bfs.h:15-36 and pagerank.h:15-29, used to cross-check the GPU results through the same API
template <typename T>
static std::vector<uint32_t> host_bfs(T &graph, uint32_t start) {
  uint64_t n = graph.get_n();
  std::vector<uint32_t> out(n, UINT32_MAX), queue{start};
  out[start] = 0;
  for (size_t h = 0; h < queue.size(); h++)
    for (int nb : graph.get_neighbourhood(queue[h]))
      if (out[nb] == UINT32_MAX) { out[nb] = out[queue[h]] + 1; queue.push_back(nb); }
  return out;
}

static void run(bool lock_search) {
  {  // Initialization
    PPPCSR pcsr(10, 10, lock_search, 1, 1, false);
    EXPECT_EQ(pcsr.get_n(), 10u);
  }
  {  // add_node
    PPPCSR pcsr(0, 0, lock_search, 1, 1, false);
    EXPECT_EQ(pcsr.get_n(), 0u);
    pcsr.add_node();
    EXPECT_EQ(pcsr.get_n(), 1u);
    EXPECT_EQ(pcsr.get_neighbourhood(0).size(), 0u);
  }
  {  // add_edge
    PPPCSR pcsr(10, 10, lock_search, 1, 1, false);
    pcsr.add_edge(11, 1, 1);  // no such node: ignored
    pcsr.add_edge(0, 1, 1);
    EXPECT_TRUE(pcsr.edge_exists(0, 1));
    EXPECT_EQ(pcsr.get_neighbourhood(0).size(), 1u);
    EXPECT_EQ(pcsr.get_n(), 10u);
    EXPECT_EQ(pcsr.get_neighbourhood(2).size(), 0u);
  }
  {  // remove_edge
    PPPCSR pcsr(10, 10, lock_search, 1, 1, false);
    pcsr.add_node();
    pcsr.remove_edge(0, 1);
    EXPECT_FALSE(pcsr.edge_exists(0, 1));
    pcsr.add_edge(0, 1, 1);
    EXPECT_TRUE(pcsr.edge_exists(0, 1));
    EXPECT_EQ(pcsr.get_neighbourhood(0).size(), 1u);
    pcsr.remove_edge(0, 1);
    EXPECT_FALSE(pcsr.edge_exists(0, 1));
    EXPECT_EQ(pcsr.get_neighbourhood(2).size(), 0u);
  }
  {  // add_remove_edge_1E4_seq (edge_exists sampled every 97th op: each check is a device round trip)
    PCSR pcsr(10, 10, lock_search, 0);
    const int edge_count = 10000;
    for (int i = 1; i < edge_count + 1; ++i) {
      pcsr.add_edge(0, i, i);
      if (i % 97 == 0) {
        EXPECT_TRUE(pcsr.edge_exists(0, i));
        for (uint32_t j = 0; j < pcsr.edges.N / pcsr.edges.logN; ++j) EXPECT_TRUE(pcsr.edges.node_locks[j]->lockable());
        EXPECT_TRUE(pcsr.edges.global_lock->lockable());
      }
    }
    EXPECT_EQ(pcsr.get_n(), 10u);
    EXPECT_EQ(pcsr.getNode(0).num_neighbors, (uint32_t)edge_count);
    for (int i = 1; i < edge_count + 1; ++i) {
      pcsr.remove_edge(0, i);
      if (i % 97 == 0) EXPECT_FALSE(pcsr.edge_exists(0, i));
    }
    EXPECT_EQ(pcsr.get_neighbourhood(0).size(), 0u);
    EXPECT_EQ(pcsr.get_n(), 10u);
  }
  {  // add_remove_edge_random_2E4_seq with a portable PRNG
    PCSR pcsr(1000, 1000, lock_search, 0);
    uint64_t x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 11); };
    for (int i = 1; i < 20001; ++i) {
      const int src = rnd() % 1000, target = rnd() % 1000;
      if (rnd() % 4 != 0) {
        pcsr.add_edge(src, target, i);
        if (i % 53 == 0) EXPECT_TRUE(pcsr.edge_exists(src, target));
      } else {
        pcsr.remove_edge(src, target);
        if (i % 53 == 0) EXPECT_FALSE(pcsr.edge_exists(src, target));
      }
    }
  }
  {  // bfs_5E4 / pagerank_5E4 (only the result size is asserted, as in the reference)
    PCSR pcsr(1000, 1000, lock_search, 0);
    uint64_t x = 1234567ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 11); };
    std::vector<ppcsr_op> adds;
    for (int i = 1; i < 50001; ++i) {
      const uint32_t s = rnd() % 1000, d = rnd() % 1000;
      adds.push_back(ppcsr_op{s, d, (uint32_t)i});
    }
    for (const auto &a : adds) pcsr.add_edge(a.src, a.dst, a.op);
    auto res = bfs(pcsr, 0);  // host/bfs.h -> ppcsr_bfs on the GPU
    EXPECT_EQ(res.size(), 1000u);
    EXPECT_TRUE(res == host_bfs(pcsr, 0));
    std::vector<float> weights(pcsr.get_n(), 1.0f), output(pcsr.get_n(), 0.0f);
    for (uint64_t i = 0; i < pcsr.get_n(); i++) {  // src/utility/pagerank.h:15-29 on the host
      const float contrib = weights[i] / pcsr.getNode((int)i).num_neighbors;
      for (int nb : pcsr.get_neighbourhood((int)i)) output[nb] += contrib;
    }
    auto pr = pagerank(pcsr, weights);  // host/pagerank.h -> ppcsr_pagerank on the GPU
    EXPECT_EQ(pr.size(), 1000u);
    EXPECT_TRUE(memcmp(pr.data(), output.data(), output.size() * sizeof(float)) == 0);
    // the non-parity bulk build of the same adds gives the same graph (neighbourhoods, BFS levels, PageRank bits)
    PCSR bulk(1000, 1000, lock_search, 0);
    bulk.bulk_build(adds);
    EXPECT_TRUE(bfs(bulk, 0) == res);
    EXPECT_TRUE(bulk.get_neighbourhood(7) == pcsr.get_neighbourhood(7));
    auto pr2 = pagerank(bulk, weights);
    EXPECT_TRUE(memcmp(pr2.data(), pr.data(), pr.size() * sizeof(float)) == 0);
    EXPECT_EQ(output.size(), 1000u);
  }
}

// the host mirror keeps submitted updates in memory until a reader or flush() applies them; past the high-water mark
// (PPCSR_PENDING_MAX, here 1000) the submitter applies the backlog itself, so a writer-only client stays bounded
static void high_water() {
  PCSR pcsr(100, 100, true, 0);
  for (uint32_t i = 0; i < 5000; ++i) {
    pcsr.add_edge(i % 100, (i * 7) % 100, 1 + i);
    EXPECT_TRUE(pcsr.pending() < 1000u);
  }
  PPPCSR pp(100, 100, true, 1, 4, false);
  for (uint32_t i = 0; i < 5000; ++i) pp.add_edge(i % 100, (i * 7) % 100, 1 + i);
  for (uint32_t i = 0; i < 5000; i += 37) EXPECT_TRUE(pp.edge_exists(i % 100, (i * 7) % 100) == pcsr.edge_exists(i % 100, (i * 7) % 100));
  EXPECT_TRUE(pp.get_neighbourhood(3) == pcsr.get_neighbourhood(3));
}

int main() {
  PCSR::quiet() = true;
  setenv("PPCSR_PENDING_MAX", "1000", 1);  // (read once, at the first submit)
  high_water();
  run(false);
  run(true);
  std::printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
  return failures ? 1 : 0;
}
