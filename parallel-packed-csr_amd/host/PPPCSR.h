// Host-side mirror of the reference class PPPCSR (reference: src/pppcsr/PPPCSR.h:11-60, PPPCSR.cpp:13-80):
// vertex-range partitioning over independent PCSRs, one partition per GPU (round-robin over `devices`).
#ifndef PPCSR_HOST_PPPCSR_H
#define PPCSR_HOST_PPPCSR_H
#include <cmath>
#include <iostream>
#include <memory>
#include <vector>

#include "PCSR.h"

class PPPCSR {
 public:
  edge_list_t edges;  // unused, kept for source compatibility (reference PPPCSR.h:14)

  // reference: PPPCSR(init_n, src_n, lock_search, numDomain, partitionsPerDomain, use_numa)   PPPCSR.cpp:13
  // use_numa placed partitions on NUMA domains; here partition p is placed on devices[p % devices.size()]
  PPPCSR(uint32_t init_n, uint32_t src_n, bool lock_search, int numDomain, int partitionsPerDomain, bool use_numa,
         std::vector<int> devices = {0})
      : partitionsPerDomain(partitionsPerDomain) {
    (void)src_n;
    (void)use_numa;
    const std::size_t P = (std::size_t)numDomain * (std::size_t)partitionsPerDomain;
    partitions.reserve(P);
    distribution.reserve(P);
    distribution.push_back(0);
    std::size_t partitionSize = init_n / P;  // PPPCSR.cpp:20 (the std::ceil there wraps an integer division)
    for (std::size_t k = 0; k < P; k++) {
      if (k > 0) distribution.push_back(distribution.back() + partitionSize);
      std::size_t size = partitionSize;
      if (k == P - 1) size = init_n - k * partitionSize;  // the last partition takes the remainder (PPPCSR.cpp:27-29)
      partitions.emplace_back(new PCSR((uint32_t)size, (uint32_t)size, lock_search, devices[k % devices.size()]));
    }
    if (!PCSR::quiet()) std::cout << "Number of partitions: " << partitions.size() << std::endl;
  }

  bool edge_exists(uint32_t src, uint32_t dest) { auto p = get_partiton(src); return partitions[p]->edge_exists(src - distribution[p], dest); }
  void add_node() { partitions.back()->add_node(); }  // PPPCSR.cpp:44
  void add_edge(uint32_t src, uint32_t dest, uint32_t value) { auto p = get_partiton(src); partitions[p]->add_edge(src - distribution[p], dest, value); }
  void remove_edge(uint32_t src, uint32_t dest) { auto p = get_partiton(src); partitions[p]->remove_edge(src - distribution[p], dest); }
  void read_neighbourhood(int src) { auto p = get_partiton(src); partitions[p]->read_neighbourhood(src - (int)distribution[p]); }
  std::vector<int> get_neighbourhood(int src) { auto p = get_partiton(src); return partitions[p]->get_neighbourhood(src - (int)distribution[p]); }

  std::size_t get_partiton(size_t vertex_id) const {  // (sic) PPPCSR.cpp:58-66
    for (std::size_t i = 1; i < distribution.size(); i++)
      if (distribution[i] > vertex_id) return i - 1;
    return distribution.size() - 1;
  }
  uint64_t get_n() {
    uint64_t n = 0;
    for (auto &p : partitions) n += p->get_n();
    return n;
  }
  node_t getNode(int id) { auto p = get_partiton(id); return partitions[p]->getNode(id - (int)distribution[p]); }
  void registerThread(int par) { partitions[par]->edges.global_lock->registerThread(); }
  void unregisterThread(int par) { partitions[par]->edges.global_lock->unregisterThread(); }

  void flush() { for (auto &p : partitions) p->flush(); }
  PCSR &partition(std::size_t k) { return *partitions[k]; }
  std::size_t num_partitions() const { return partitions.size(); }

 private:
  std::vector<std::unique_ptr<PCSR>> partitions;
  std::vector<size_t> distribution;
  int partitionsPerDomain;
};

#endif  // PPCSR_HOST_PPPCSR_H
