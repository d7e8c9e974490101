// Host-side mirror of the reference class PPPCSR (reference: src/pppcsr/PPPCSR.h:11-60, PPPCSR.cpp:13-80) over the
// pppcsr_* entry points of the C ABI: vertex-range partitioning over independent PCSRs, the partitions of domain d on
// GPU devices[d].  Writers enqueue into ONE batch in arrival order (global vertex ids); flush() hands it to
// pppcsr_apply_batch, which buckets it by owner (stable, src made partition-local, PPPCSR.cpp:46-52) and drives every
// partition from its own host thread and stream — all domains at once, as the reference's ThreadPoolPPPCSR does
// (thread_pool_pppcsr.cpp:121-156).  Thread safety as in PCSR.h.
#ifndef PPCSR_HOST_PPPCSR_H
#define PPCSR_HOST_PPPCSR_H
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <mutex>
#include <vector>

#include "PCSR.h"

class PPPCSR {
 public:
  edge_list_t edges;  // unused, kept for source compatibility (reference PPPCSR.h:14)

  // reference: PPPCSR(init_n, src_n, lock_search, numDomain, partitionsPerDomain, use_numa)   PPPCSR.cpp:13
  // use_numa placed the partitions of a domain on its NUMA node; here domain d is GPU devices[d % devices.size()]
  PPPCSR(uint32_t init_n, uint32_t src_n, bool lock_search, int numDomain, int partitionsPerDomain, bool use_numa,
         std::vector<int> devices = {0})
      : partitionsPerDomain(partitionsPerDomain) {
    (void)use_numa;
    check(pppcsr_create(init_n, src_n, lock_search ? 1 : 0, numDomain, partitionsPerDomain, devices.data(), (int)devices.size(), &h_));
    uint64_t P = 0;
    check(pppcsr_num_partitions(h_, &P));
    for (uint64_t k = 0; k < P; k++) {
      uint64_t first = 0;
      ppcsr_t ph = nullptr;
      check(pppcsr_partition_start(h_, k, &first));
      check(pppcsr_partition(h_, k, &ph));
      distribution.push_back((size_t)first);
      partitions.emplace_back(new PCSR(PCSR::Borrowed{}, ph));
    }
    if (!PCSR::quiet()) std::cout << "Number of partitions: " << partitions.size() << std::endl;
  }
  ~PPPCSR() {
    partitions.clear();  // (views only)
    pppcsr_destroy(h_);
  }
  PPPCSR(const PPPCSR &) = delete;
  PPPCSR &operator=(const PPPCSR &) = delete;

  // readers hold the engine mutex across flush + query: a partition's engine must never be entered while
  // pppcsr_apply_batch drives it from a worker thread
  bool edge_exists(uint32_t src, uint32_t dest) {
    std::lock_guard<std::mutex> g(engine_mu_);
    flush_locked();
    auto p = get_partiton(src);
    return partitions[p]->edge_exists(src - distribution[p], dest);
  }
  void add_node() {  // PPPCSR.cpp:44
    std::lock_guard<std::mutex> g(engine_mu_);
    flush_locked();
    partitions.back()->add_node();
  }
  void add_edge(uint32_t src, uint32_t dest, uint32_t value) {  // PPPCSR.cpp:46-48
    if (value == 0) return;
    enqueue(ppcsr_op{src, dest, value});
  }
  void remove_edge(uint32_t src, uint32_t dest) {  // PPPCSR.cpp:50-52
    enqueue(ppcsr_op{src, dest, 0u});
  }
  void read_neighbourhood(int src) {
    std::lock_guard<std::mutex> g(engine_mu_);
    flush_locked();
    auto p = get_partiton(src);
    partitions[p]->read_neighbourhood(src - (int)distribution[p]);
  }
  std::vector<int> get_neighbourhood(int src) {
    std::lock_guard<std::mutex> g(engine_mu_);
    flush_locked();
    auto p = get_partiton(src);
    return partitions[p]->get_neighbourhood(src - (int)distribution[p]);
  }

  std::size_t get_partiton(size_t vertex_id) const {  // (sic) PPPCSR.cpp:58-66
    for (std::size_t i = 1; i < distribution.size(); i++)
      if (distribution[i] > vertex_id) return i - 1;
    return distribution.size() - 1;
  }
  uint64_t get_n() {
    std::lock_guard<std::mutex> g(engine_mu_);
    uint64_t n = 0;
    for (auto &p : partitions) n += p->get_n();
    return n;
  }
  node_t getNode(int id) {
    std::lock_guard<std::mutex> g(engine_mu_);
    flush_locked();
    auto p = get_partiton(id);
    return partitions[p]->getNode(id - (int)distribution[p]);
  }
  void registerThread(int par) { partitions[par]->edges.global_lock->registerThread(); }
  void unregisterThread(int par) { partitions[par]->edges.global_lock->unregisterThread(); }

  // apply everything enqueued so far: ONE pppcsr_apply_batch call = stable owner bucketing + all partitions concurrently
  void flush() {
    std::lock_guard<std::mutex> g(engine_mu_);
    flush_locked();
  }
  // a partition's view; the caller must not use it while other threads write to this PPPCSR
  PCSR &partition(std::size_t k) { flush(); return *partitions[k]; }
  std::size_t num_partitions() const { return partitions.size(); }
  pppcsr_t handle() { return h_; }

 private:
  void enqueue(const ppcsr_op &op) {  // (bounded like PCSR::enqueue: past the high-water mark the submitter applies the backlog)
    std::size_t held;
    {
      std::lock_guard<std::mutex> g(pending_mu_);
      pending_.push_back(op);
      held = pending_.size();
    }
    if (held >= PCSR::pending_high_water()) flush();
  }
  void flush_locked() {  // engine_mu_ is held
    std::vector<ppcsr_op> batch;
    {
      std::lock_guard<std::mutex> g2(pending_mu_);
      if (pending_.empty()) return;
      batch.swap(pending_);
    }
    const bool timing = std::getenv("PPCSR_CLI_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    for (auto &p : partitions) p->edges.global_lock->applying.store(1, std::memory_order_release);  // (FastLock::lockable() of the views)
    const int rc = pppcsr_apply_batch(h_, batch.data(), batch.size());
    for (auto &p : partitions) p->edges.global_lock->applying.store(0, std::memory_order_release);
    check(rc);
    const auto t1 = std::chrono::steady_clock::now();
    for (auto &p : partitions) p->sync_geometry();
    if (timing)
      std::cerr << "[ppcsr_cli] pppcsr_apply_batch(" << batch.size() << ") " << std::chrono::duration<double, std::milli>(t1 - t0).count()
                << " ms, geometry refresh " << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count() << " ms" << std::endl;
  }
  static void check(int rc) {
    if (rc != 0) {  // the reference exits on failure (PCSR.cpp:49-54)
      std::cout << "ppcsr: " << ppcsr_strerror(rc) << ": " << ppcsr_last_error() << std::endl;
      std::exit(EXIT_FAILURE);
    }
  }
  pppcsr_t h_ = nullptr;
  std::vector<std::unique_ptr<PCSR>> partitions;  // views of the partitions' engines (owned by h_)
  std::vector<size_t> distribution;
  std::mutex pending_mu_, engine_mu_;
  std::vector<ppcsr_op> pending_;
  int partitionsPerDomain;
};

#endif  // PPCSR_HOST_PPPCSR_H
