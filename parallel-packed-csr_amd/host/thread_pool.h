// Pool shims with the reference's duck-typed interface (src/main.cpp:65-83 ThreadPool_t concept; reference classes
// src/thread_pool/thread_pool.h:17-40 and src/thread_pool_pppcsr/thread_pool_pppcsr.h:17-47).
// submit_* only enqueue; start() applies everything submitted since the last start() as ONE batch on the GPU(s) with
// sequential stream-order semantics; stop() prints the reference's "Elapsed wall clock time: <ms>" line.
#ifndef PPCSR_HOST_THREAD_POOL_H
#define PPCSR_HOST_THREAD_POOL_H
#include <chrono>
#include <iostream>
#include <vector>

#include "PPPCSR.h"

template <class Graph>
class PoolShim {
 public:
  Graph *pcsr = nullptr;
  void submit_add(int thread_id, int src, int dest) { count(thread_id); pcsr->add_edge((uint32_t)src, (uint32_t)dest, 1); }
  void submit_delete(int thread_id, int src, int dest) { count(thread_id); pcsr->remove_edge((uint32_t)src, (uint32_t)dest); }
  void submit_read(int thread_id, int src) { count(thread_id); reads_.push_back(src); }
  void start(int threads) {
    s_ = std::chrono::steady_clock::now();
    started_threads_ = threads;
    for (int i = 0; i < threads; i++)
      std::cout << "Thread " << i << " has " << (i < (int)tasks_.size() ? tasks_[i] : 0) << " tasks" << std::endl;
    pcsr->flush();
    for (int r : reads_) pcsr->read_neighbourhood(r);
    reads_.clear();
    tasks_.assign(tasks_.size(), 0);
  }
  void stop() {
    for (int i = 1; i < started_threads_; i++) std::cout << "Done" << std::endl;
    const auto e = std::chrono::steady_clock::now();
    std::cout << "Elapsed wall clock time: " << std::chrono::duration_cast<std::chrono::milliseconds>(e - s_).count() << std::endl;
  }
  ~PoolShim() { delete pcsr; }

 protected:
  void count(int thread_id) {
    if (thread_id >= (int)tasks_.size()) tasks_.resize(thread_id + 1, 0);
    if (thread_id >= 0) tasks_[thread_id]++;
  }
  std::vector<size_t> tasks_;
  std::vector<int> reads_;
  std::chrono::steady_clock::time_point s_;
  int started_threads_ = 0;
};

class ThreadPool : public PoolShim<PCSR> {
 public:
  // reference: ThreadPool(NUM_OF_THREADS, lock_search, init_num_nodes, partitions_per_domain)   thread_pool.cpp:19-23
  ThreadPool(const int NUM_OF_THREADS, bool lock_search, uint32_t init_num_nodes, int partitions_per_domain, int device = 0) {
    (void)partitions_per_domain;
    tasks_.resize(NUM_OF_THREADS, 0);
    pcsr = new PCSR(init_num_nodes, init_num_nodes, lock_search, device);
  }
};

class ThreadPoolPPPCSR : public PoolShim<PPPCSR> {
 public:
  // reference: ThreadPoolPPPCSR(NUM_OF_THREADS, lock_search, init_num_nodes, partitions_per_domain, use_numa)
  // (thread_pool_pppcsr.cpp:20-48); the NUMA-domain count of the reference becomes the number of GPUs
  ThreadPoolPPPCSR(const int NUM_OF_THREADS, bool lock_search, uint32_t init_num_nodes, int partitions_per_domain, bool use_numa,
                   std::vector<int> devices = {0}) {
    tasks_.resize(NUM_OF_THREADS, 0);
    pcsr = new PPPCSR(init_num_nodes, init_num_nodes, lock_search, (int)devices.size(), partitions_per_domain, use_numa, devices);
  }
};

#endif  // PPCSR_HOST_THREAD_POOL_H
