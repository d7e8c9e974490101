// Drop-in for the reference's src/utility/pagerank.h (template pagerank(graph, node_values), pagerank.h:15-29): one push
// step, output[d] += node_values[s] / num_neighbors(s) over the edges (s, d) in ascending s.  For the engine-backed PCSR
// with float weights it runs on the GPU (ppcsr_pagerank; same order of fp32 additions, bit-identical); everything else
// takes the host loop.
#ifndef PPCSR_HOST_PAGERANK_H
#define PPCSR_HOST_PAGERANK_H
#include <cstdint>
#include <vector>

#include "PCSR.h"

inline std::vector<float> pagerank(PCSR &graph, const std::vector<float> &node_values) { return graph.pagerank(node_values); }

template <typename T, typename weight_t>
std::vector<weight_t> pagerank(T &graph, const std::vector<weight_t> &node_values) {
  const uint64_t n = graph.get_n();
  std::vector<weight_t> output(n, 0);
  for (uint64_t s = 0; s < n; s++) {
    const weight_t share = node_values[s] / graph.getNode((int)s).num_neighbors;
    for (const int d : graph.get_neighbourhood((int)s)) output[d] += share;
  }
  return output;
}
#endif
