// Host-side mirror of the reference class PCSR (reference: src/pcsr/PCSR.h:64-202) over the MI355X engine's C ABI.
// Same type names, method names, argument meaning and observable behaviour as the reference so that code written
// against the reference (tests, bfs.h / pagerank.h templates, thread pools, main.cpp) compiles against this header.
// The state lives in HBM; writers enqueue into a batch that is applied — with the reference's sequential
// stream-order semantics — at the next read or at flush() (never from a submit path: everything a pool has been given
// before start() is applied between start() and stop(), as the reference's timing protocol requires).
//
// Threading (reference contract: any number of threads may call add_edge / remove_edge after registerThread(),
// thread_pool.cpp:39-48, DataStructureTest.cpp:81-120): writers append to the pending batch under a mutex, so each
// thread's own updates keep their program order and updates of different threads interleave in arrival order — the
// nondeterminism the reference has too (its final array depends on the thread schedule).  Engine calls
// (flush and every reader) are serialised by a second mutex; readers therefore see every update that was submitted
// before their call.
#ifndef PPCSR_HOST_PCSR_H
#define PPCSR_HOST_PCSR_H
#include <ppcsr.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <mutex>
#include <vector>

typedef struct _node {  // reference PCSR.h:18-23
  uint32_t beginning;
  uint32_t end;
  uint32_t num_neighbors;
} node_t;
typedef struct _edge {  // reference PCSR.h:30-35
  uint32_t src;
  uint32_t dest;
  uint32_t value;
} edge_t;

// The reference exposes its locks through `edges` (PCSR.h:37-44) and its tests poke them
// (DataStructureTest.cpp:58-62, 86-92).  On the GPU conflict-free rounds replace leaf locks, so there is nothing to hold
// between calls: FastLock keeps the registered-thread count of the reference's cooperative lock (fastLock.h:26-51) and
// reports lockable() whenever no batch is being applied; the per-leaf HybridLock is always free.
struct FastLock {
  std::atomic<int> registered{0};
  std::atomic<int> applying{0};
  void registerThread() { registered.fetch_add(1, std::memory_order_relaxed); }
  void unregisterThread() { registered.fetch_sub(1, std::memory_order_relaxed); }
  bool lockable() { return applying.load(std::memory_order_acquire) == 0; }
};
struct HybridLock {
  bool lockable() { return true; }
};
typedef struct edge_list {
  uint64_t N = 0;
  int H = 0;
  int logN = 0;
  std::shared_ptr<FastLock> global_lock = std::make_shared<FastLock>();
  HybridLock **node_locks = nullptr;  // node_locks[i] is valid for i < N / logN
  edge_t *items = nullptr;            // host mirror, valid after PCSR::download()
} edge_list_t;

class PCSR {
 public:
  edge_list_t edges;  // N / logN / H are refreshed after every flush

  // reference: PCSR(uint32_t init_n, uint32_t src_n, bool lock_search, int domain = 0)   PCSR.cpp:775
  // `domain` selected the NUMA node; here it selects the GPU (negative = device 0)
  PCSR(uint32_t init_n, uint32_t src_n, bool lock_search, int domain = 0) {
    check(ppcsr_create(init_n, src_n, lock_search ? 1 : 0, domain < 0 ? 0 : domain, &h_));
    refresh_geometry(true);
  }
  // view of an engine owned elsewhere (a partition of a pppcsr_t): same methods, no ownership
  struct Borrowed {};
  PCSR(Borrowed, ppcsr_t h) : h_(h), owns_(false) { refresh_geometry(true, true); }
  ~PCSR() {
    if (owns_) ppcsr_destroy(h_);
    delete[] lock_store_;
    delete[] lock_ptrs_;
  }
  PCSR(const PCSR &) = delete;
  PCSR &operator=(const PCSR &) = delete;

  /** Public API (reference PCSR.h:72-124) */
  bool edge_exists(uint32_t src, uint32_t dest) {  // PCSR.cpp:860
    std::lock_guard<std::mutex> g(engine_mu_);
    flush_locked();
    int e = 0;
    check(ppcsr_edge_exists(h_, src, dest, &e));
    return e != 0;
  }
  void add_node() {  // PCSR.cpp:681
    std::lock_guard<std::mutex> g(engine_mu_);
    flush_locked();
    check(ppcsr_add_node(h_));
    refresh_geometry(false);
  }
  void add_edge(uint32_t src, uint32_t dest, uint32_t value) {  // PCSR.cpp:706 (value 0 / src >= n: silently ignored)
    if (value == 0) return;
    enqueue(ppcsr_op{src, dest, value});
  }
  void remove_edge(uint32_t src, uint32_t dest) {  // PCSR.cpp:709
    enqueue(ppcsr_op{src, dest, 0u});
  }
  // Updates wait in host memory (12 B each) until a reader or flush() applies them.  A writer-only client would let that
  // grow with the stream, so past a high-water mark (default 64 Mi updates = 768 MiB; PPCSR_PENDING_MAX overrides) the
  // submitting thread applies what is there.  The reference's benchmark batches (<= 10 M updates) never reach it, so no
  // engine call lands inside a pool's timed submit window.
  static std::size_t pending_high_water() {
    static const std::size_t v = [] {
      const char *e = std::getenv("PPCSR_PENDING_MAX");
      const unsigned long long x = e ? std::strtoull(e, nullptr, 10) : 0ull;
      return x ? (std::size_t)x : ((std::size_t)64 << 20);
    }();
    return v;
  }
  void read_neighbourhood(int src) {  // PCSR.cpp:892
    std::lock_guard<std::mutex> g(engine_mu_);
    flush_locked();
    check(ppcsr_read_neighbourhood(h_, src));
  }
  std::vector<int> get_neighbourhood(int src) {  // PCSR.cpp:901
    std::lock_guard<std::mutex> g(engine_mu_);
    flush_locked();
    uint64_t c = 0;
    check(ppcsr_get_neighbourhood(h_, src, nullptr, 0, &c));
    std::vector<int> out(c);
    if (c) check(ppcsr_get_neighbourhood(h_, src, out.data(), c, &c));
    return out;
  }
  uint64_t get_n() {  // PCSR.cpp:100
    std::lock_guard<std::mutex> g(engine_mu_);
    return get_n_locked();
  }
  // reference returns node_t& into host memory (PCSR.h:118); the state is in HBM, so this is a copy
  node_t getNode(int id) {
    std::lock_guard<std::mutex> g(engine_mu_);
    flush_locked();
    ppcsr_node nd;
    check(ppcsr_get_node(h_, (uint32_t)id, &nd));
    return node_t{nd.beginning, nd.end, nd.num_neighbors};
  }

  /** additions that have no reference equivalent */
  void flush() {  // apply everything enqueued so far, in arrival order
    std::lock_guard<std::mutex> g(engine_mu_);
    flush_locked();
  }
  void sync_geometry() {  // after the engine was driven through another handle (PPPCSR applies through pppcsr_apply_batch)
    std::lock_guard<std::mutex> g(engine_mu_);
    refresh_geometry(false);
  }
  void download(std::vector<edge_t> *items, std::vector<node_t> *nodes) {  // raw state, reference byte layout
    std::lock_guard<std::mutex> g(engine_mu_);
    flush_locked();
    items->resize(edges.N);
    nodes->resize(get_n_locked());
    check(ppcsr_export_state(h_, reinterpret_cast<ppcsr_edge *>(items->data()), reinterpret_cast<ppcsr_node *>(nodes->data())));
  }
  // the reference's consumers (src/utility/bfs.h, pagerank.h) run on the device over the gapped array; host/bfs.h and
  // host/pagerank.h route the reference's free functions here
  // NON-parity fast path (no reference equivalent): build an empty graph from a list of adds in a few device passes;
  // same neighbourhoods / values / num_neighbors / invariants as inserting them one by one, different slot layout
  void bulk_build(const std::vector<ppcsr_op> &adds) {
    std::lock_guard<std::mutex> g(engine_mu_);
    flush_locked();
    check(ppcsr_bulk_build(h_, adds.data(), adds.size(), nullptr));
    refresh_geometry(false);
  }
  std::vector<uint32_t> bfs(uint32_t start_node) {
    std::lock_guard<std::mutex> g(engine_mu_);
    flush_locked();
    std::vector<uint32_t> out(get_n_locked());
    check(ppcsr_bfs(h_, start_node, out.data(), nullptr));
    return out;
  }
  std::vector<float> pagerank(const std::vector<float> &node_values) {
    std::lock_guard<std::mutex> g(engine_mu_);
    flush_locked();
    std::vector<float> out(get_n_locked());
    check(ppcsr_pagerank(h_, node_values.data(), out.data(), nullptr));
    return out;
  }
  ppcsr_t handle() { return h_; }
  size_t pending() {
    std::lock_guard<std::mutex> g(pending_mu_);
    return pending_.size();
  }
  static bool &quiet() {
    static bool q = false;
    return q;
  }

 private:
  // engine_mu_ is held
  void enqueue(const ppcsr_op &op) {
    std::size_t held;
    {
      std::lock_guard<std::mutex> g(pending_mu_);
      pending_.push_back(op);
      held = pending_.size();
    }
    if (held >= pending_high_water()) flush();
  }
  void flush_locked() {
    std::vector<ppcsr_op> batch;
    {
      std::lock_guard<std::mutex> g(pending_mu_);
      if (pending_.empty()) return;
      batch.swap(pending_);
    }
    edges.global_lock->applying.store(1, std::memory_order_release);
    const int rc = ppcsr_apply_batch(h_, batch.data(), batch.size());
    edges.global_lock->applying.store(0, std::memory_order_release);  // (before check(): it may not return)
    check(rc);
    refresh_geometry(false);
  }
  uint64_t get_n_locked() {
    uint64_t n = 0;
    check(ppcsr_get_n(h_, &n));
    return n;
  }
  static void check(int rc) {
    if (rc != 0) {  // the reference exits on failure (PCSR.cpp:49-54); keep that behaviour at this level
      std::cout << "ppcsr: " << ppcsr_strerror(rc) << ": " << ppcsr_last_error() << std::endl;
      std::exit(EXIT_FAILURE);
    }
  }
  void refresh_geometry(bool first, bool silent = false) {
    uint64_t N = 0;
    int lg = 0, H = 0;
    check(ppcsr_geometry(h_, &N, &lg, &H));
    if (first || N != edges.N) {
      edges.N = N;
      edges.logN = lg;
      edges.H = H;
      // same line the reference prints from resizeEdgeArray (PCSR.cpp:72); intermediate sizes inside one batch are not shown
      if (!quiet() && !silent) std::cout << "Edges: " << N << " logN: " << lg << " #count: " << N / lg << std::endl;
      const uint64_t nl = N / (uint64_t)lg;
      if (nl > lock_cap_) {
        delete[] lock_store_;
        delete[] lock_ptrs_;
        lock_store_ = new HybridLock[nl];
        lock_ptrs_ = new HybridLock *[nl];
        for (uint64_t i = 0; i < nl; i++) lock_ptrs_[i] = &lock_store_[i];
        lock_cap_ = nl;
      }
      edges.node_locks = lock_ptrs_;
    }
  }

  ppcsr_t h_ = nullptr;
  bool owns_ = true;
  std::mutex pending_mu_;  // guards pending_ (writers)
  std::mutex engine_mu_;   // serialises engine calls (flush + readers); taken before pending_mu_
  std::vector<ppcsr_op> pending_;
  HybridLock *lock_store_ = nullptr;
  HybridLock **lock_ptrs_ = nullptr;
  uint64_t lock_cap_ = 0;
};

#endif  // PPCSR_HOST_PCSR_H
