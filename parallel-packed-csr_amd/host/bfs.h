// Drop-in for the reference's src/utility/bfs.h (template bfs(graph, start_node), bfs.h:15-36).
// For the engine-backed PCSR the walk runs on the GPU over the gapped array (ppcsr_bfs); any other graph type gets the
// reference's host algorithm: a queue-based walk through get_neighbourhood().
#ifndef PPCSR_HOST_BFS_H
#define PPCSR_HOST_BFS_H
#include <cstdint>
#include <vector>

#include "PCSR.h"

inline std::vector<uint32_t> bfs(PCSR &graph, uint32_t start_node) { return graph.bfs(start_node); }

template <typename T>
std::vector<uint32_t> bfs(T &graph, uint32_t start_node) {
  const uint64_t n = graph.get_n();
  std::vector<uint32_t> level(n, UINT32_MAX), order{start_node};
  level[start_node] = 0;
  for (size_t head = 0; head < order.size(); head++) {
    const uint32_t u = order[head];
    for (const int nb : graph.get_neighbourhood((int)u))
      if (level[nb] == UINT32_MAX) {
        level[nb] = level[u] + 1;
        order.push_back((uint32_t)nb);
      }
  }
  return level;
}
#endif
