// C ABI (include/ppcsr.h) over ppcsr::Engine.  No torch types, plain pointers and sizes.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ppcsr.h"
#include "engine.h"

using ppcsr::Engine;

// small device helpers implemented in engine.cc (this file stays free of runtime headers)
int capi_set_device(int d);
int capi_dev_alloc(void **p, size_t bytes);
int capi_dev_free(void *p);
int capi_d2h_sync(void *dst, const void *src, size_t bytes);  // on the NULL stream, after whatever it holds
int capi_h2d_sync(void *dst, const void *src, size_t bytes);
int capi_dev_memset(void *p, int byte, size_t bytes, void *stream);
// the carrier of the owner exchange: RCCL in the product (engine.cc); tests/hostsim provides a shared-memory one for the emulator
struct ppcsr_xchg;
int capi_xchg_unique_id(void *out128, std::string *err);
int capi_xchg_create(const void *id128, int nranks, int rank, int device, ppcsr_xchg **out, std::string *err);
int capi_xchg_destroy(ppcsr_xchg *x);
int capi_xchg_ranks(ppcsr_xchg *x, int *nranks, int *rank, int *device, void **stream);
int capi_xchg_sendrecv(ppcsr_xchg *x, uint64_t nseg, const void *const *sptr, const uint64_t *sbytes, const int *speer, void *const *rptr,
                       const uint64_t *rbytes, const int *rpeer);
const char *capi_xchg_error(ppcsr_xchg *x);

static thread_local std::string g_last_error;

struct ppcsr_engine {
  Engine *e;
};

static int ret(Engine *e, int rc) {
  if (rc != 0 && e) g_last_error = e->last_error();
  return rc;
}
static int bad(const char *msg) {
  g_last_error = msg;
  return PPCSR_STATUS_EINVAL;
}

static_assert(sizeof(ppcsr_edge) == sizeof(ppcsr::Edge), "layout");
static_assert(sizeof(ppcsr_node) == sizeof(ppcsr::Node), "layout");
static_assert(sizeof(ppcsr_op) == sizeof(ppcsr::Op), "layout");

extern "C" {

int ppcsr_device_count(void) {
  int n = 0;
  if (gpu_device_count_for_capi(&n) != 0) return 0;
  return n;
}

int ppcsr_create(uint32_t init_n, uint32_t src_n, int lock_search, int device, ppcsr_t *out) {
  if (!out) return bad("null out");
  *out = nullptr;
  Engine *e = nullptr;
  std::string msg;
  int rc = Engine::create(init_n, src_n, lock_search, device, &e, &msg);
  if (rc != 0) {
    g_last_error = msg;
    return rc;
  }
  *out = new ppcsr_engine{e};
  return 0;
}
int ppcsr_destroy(ppcsr_t h) {
  if (!h) return 0;
  delete h->e;
  delete h;
  return 0;
}
#define H_CHECK() \
  if (!h || !h->e) return bad("null handle")

int ppcsr_add_edge(ppcsr_t h, uint32_t s, uint32_t d, uint32_t v) { H_CHECK(); return ret(h->e, h->e->add_edge(s, d, v)); }
int ppcsr_remove_edge(ppcsr_t h, uint32_t s, uint32_t d) { H_CHECK(); return ret(h->e, h->e->remove_edge(s, d)); }
int ppcsr_add_node(ppcsr_t h) { H_CHECK(); return ret(h->e, h->e->add_node()); }
int ppcsr_apply_batch(ppcsr_t h, const ppcsr_op *ops, uint64_t n) {
  H_CHECK();
  return ret(h->e, h->e->apply_batch_host(reinterpret_cast<const ppcsr::Op *>(ops), n));
}
int ppcsr_apply_batch_device(ppcsr_t h, const ppcsr_op *ops, uint64_t n) {
  H_CHECK();
  return ret(h->e, h->e->apply_batch_device(reinterpret_cast<const ppcsr::Op *>(ops), n));
}
int ppcsr_edge_exists(ppcsr_t h, uint32_t s, uint32_t d, int *exists) {
  H_CHECK();
  if (!exists) return bad("null out");
  return ret(h->e, h->e->edge_exists(s, d, exists));
}
int ppcsr_get_n(ppcsr_t h, uint64_t *n) {
  H_CHECK();
  *n = h->e->n();
  return 0;
}
int ppcsr_get_node(ppcsr_t h, uint32_t v, ppcsr_node *out) {
  H_CHECK();
  return ret(h->e, h->e->get_node(v, reinterpret_cast<ppcsr::Node *>(out)));
}
int ppcsr_geometry(ppcsr_t h, uint64_t *N, int *logN, int *H) {
  H_CHECK();
  if (N) *N = h->e->N();
  if (logN) *logN = h->e->logN();
  if (H) *H = h->e->H();
  return 0;
}
int ppcsr_get_neighbourhood(ppcsr_t h, int src, int *out, uint64_t cap, uint64_t *count) {
  H_CHECK();
  uint64_t c = 0;
  int rc = h->e->get_neighbourhood(src, out, cap, &c);
  if (count) *count = c;
  if (rc == 0 && out && c > cap) return PPCSR_STATUS_ERANGE;
  return ret(h->e, rc);
}
int ppcsr_read_neighbourhood(ppcsr_t h, int src) { H_CHECK(); return ret(h->e, h->e->read_neighbourhood(src)); }
int ppcsr_scan_all(ppcsr_t h, uint64_t *row_offsets, int *dests, uint64_t cap, uint64_t *total) {
  H_CHECK();
  return ret(h->e, h->e->scan_all(row_offsets, dests, cap, total));
}
int ppcsr_bulk_build(ppcsr_t h, const ppcsr_op *adds, uint64_t n, double *device_ms) {
  H_CHECK();
  if (!adds && n) return bad("bulk_build: null input");
  return ret(h->e, h->e->bulk_build(reinterpret_cast<const ppcsr::Op *>(adds), n, device_ms));
}
int ppcsr_bfs(ppcsr_t h, uint32_t start, uint32_t *levels, double *device_ms) {
  H_CHECK();
  if (!levels) return bad("bfs: null output");
  return ret(h->e, h->e->bfs(start, levels, device_ms));
}
int ppcsr_pagerank(ppcsr_t h, const float *node_values, float *out, double *device_ms) {
  H_CHECK();
  if (!node_values || !out) return bad("pagerank: null argument");
  return ret(h->e, h->e->pagerank(node_values, out, device_ms));
}
int ppcsr_export_state(ppcsr_t h, ppcsr_edge *items, ppcsr_node *nodes) {
  H_CHECK();
  return ret(h->e, h->e->export_state(reinterpret_cast<ppcsr::Edge *>(items), reinterpret_cast<ppcsr::Node *>(nodes)));
}
int ppcsr_stats(ppcsr_t h, ppcsr_stats_t *out) {
  H_CHECK();
  ppcsr::EngineStats s;
  int rc = h->e->stats(&s);
  if (rc != 0) return ret(h->e, rc);
  out->N = s.N; out->n = s.n; out->logN = s.logN; out->H = s.H;
  out->rounds = s.rounds; out->committed = s.committed; out->planned = s.planned;
  out->exclusive_ops = s.exclusive_ops; out->round_syncs = s.round_syncs;
  out->redistribute_calls = s.redistribute_calls; out->redistribute_slots = s.redistribute_slots;
  out->double_calls = s.double_calls; out->half_calls = s.half_calls; out->big_redistributes = s.big_redistributes; out->rollbacks = s.rollbacks;
  out->not_found = s.not_found; out->duplicates = s.duplicates; out->noops = s.noops; out->slide_slots = s.slide_slots;
  out->ops_applied = s.ops_applied; out->last_batch_ms = s.last_batch_ms; out->last_batch_h2d_ms = s.last_batch_h2d_ms;
  out->prof_plan_ms = s.prof_plan_ms; out->prof_check_ms = s.prof_check_ms; out->prof_apply_ms = s.prof_apply_ms; out->prof_compact_ms = s.prof_compact_ms;
  out->prof_launches = s.prof_launches;
  out->wasted_rounds = s.wasted_rounds;
  out->narrow = s.narrow;
  out->narrow_lost = s.narrow_lost;
  return 0;
}
int ppcsr_set_option(ppcsr_t h, const char *key, int64_t value) { H_CHECK(); return ret(h->e, h->e->set_option(key, value)); }
int ppcsr_check_invariants(ppcsr_t h, uint64_t *bad_leaves) { H_CHECK(); return ret(h->e, h->e->check_invariants(bad_leaves)); }
int ppcsr_bench_scan_all(ppcsr_t h, double *ms, uint64_t *total) { H_CHECK(); return ret(h->e, h->e->scan_all_device(ms, total)); }
int ppcsr_bench_rebalance(ppcsr_t h, uint64_t w, int iters, double *ms) { H_CHECK(); return ret(h->e, h->e->rebalance_bench(w, iters, ms)); }
int ppcsr_bench_resize(ppcsr_t h, int iters, double *double_ms, double *half_ms) {
  H_CHECK();
  if (!double_ms || !half_ms) return bad("null output");
  return ret(h->e, h->e->resize_bench(iters, double_ms, half_ms));
}
int ppcsr_snapshot(ppcsr_t h) { H_CHECK(); return ret(h->e, h->e->snapshot()); }
int ppcsr_restore(ppcsr_t h) { H_CHECK(); return ret(h->e, h->e->restore()); }
const char *ppcsr_strerror(int status) { return ppcsr::error_string(status); }
const char *ppcsr_last_error(void) { return g_last_error.c_str(); }

// ---- PPPCSR ----------------------------------------------------------------------------------------------------
}  // extern "C"

struct pppcsr_engine {
  std::vector<ppcsr_engine *> parts;   // one per partition of the GLOBAL layout; nullptr = not resident in this process
  std::vector<uint64_t> distribution;  // first vertex of each partition (PPPCSR.h:57)
  std::vector<int> device;             // device of each resident partition
  uint32_t init_n;
  uint64_t total_n = 0;  // vertices over all partitions (init_n + add_node calls)
  int lock_search = 1;
  // edges on the move of pppcsr_repartition_export
  ppcsr_op *d_moved = nullptr;
  uint64_t moved_cap = 0;
  // device-resident routing scratch of pppcsr_apply_batch_device (bucketed copy of the batch + bucket sizes)
  ppcsr_op *d_bucketed = nullptr;
  uint64_t bucketed_cap = 0;
  uint64_t *d_counts = nullptr;
  // identity of this handle: a communicator's staging (pppcsr_comm::x) is keyed on it, not on the pointer — a destroyed handle's
  // address can be handed out again by the allocator
  uint64_t gen = next_generation();
  static uint64_t next_generation() {
    static std::atomic<uint64_t> g{0};
    return ++g;
  }
};

// PPPCSR.cpp:20-29: partitionSize = floor(init_n / P) (the std::ceil wraps an integer division); last takes the rest
static void partition_layout(uint32_t init_n, uint64_t P, std::vector<uint64_t> *dist, std::vector<uint64_t> *sizes) {
  dist->assign(P, 0);
  sizes->assign(P, 0);
  const uint64_t ps = init_n / P;
  for (uint64_t k = 0; k < P; k++) {
    if (k > 0) (*dist)[k] = (*dist)[k - 1] + ps;
    (*sizes)[k] = (k == P - 1) ? (init_n - k * ps) : ps;
  }
}
// PPPCSR.cpp:58-66
static uint64_t owner_of(const std::vector<uint64_t> &dist, uint64_t v) {
  for (size_t i = 1; i < dist.size(); i++)
    if (dist[i] > v) return i - 1;
  return dist.size() - 1;
}

extern "C" {

// partitions [first, first + n_local) of the global layout are created in this process; partition k lives on the device
// of its DOMAIN (the reference allocates the partitions of domain d on NUMA node d, PPPCSR.cpp:24-31): devices[(k / ppd) % n]
static int create_parts(uint32_t init_n, int lock_search, int num_domains, int parts_per_domain, uint64_t first, uint64_t n_local,
                        const int *devices, int n_devices, pppcsr_t *out) {
  if (!out || num_domains < 1 || parts_per_domain < 1) return bad("bad partition counts");
  *out = nullptr;
  const uint64_t P = (uint64_t)num_domains * (uint64_t)parts_per_domain;
  if (first + n_local > P) return bad("local partition range exceeds the layout");
  std::unique_ptr<pppcsr_engine> pp(new pppcsr_engine());
  pp->init_n = init_n;
  pp->total_n = init_n;
  pp->lock_search = lock_search;
  std::vector<uint64_t> sizes;
  partition_layout(init_n, P, &pp->distribution, &sizes);
  pp->parts.assign(P, nullptr);
  pp->device.assign(P, 0);
  for (uint64_t k = first; k < first + n_local; k++) {
    ppcsr_t h = nullptr;
    const int dev = (devices && n_devices > 0) ? devices[(k / (uint64_t)parts_per_domain) % (uint64_t)n_devices] : 0;
    int rc = ppcsr_create((uint32_t)sizes[k], (uint32_t)sizes[k], lock_search, dev, &h);
    if (rc != 0) {
      for (auto *q : pp->parts) ppcsr_destroy(q);
      return rc;
    }
    pp->parts[k] = h;
    pp->device[k] = dev;
  }
  *out = pp.release();
  return 0;
}
int pppcsr_create(uint32_t init_n, uint32_t src_n, int lock_search, int num_domains, int parts_per_domain, const int *devices,
                  int n_devices, pppcsr_t *out) {
  (void)src_n;
  if (num_domains < 1 || parts_per_domain < 1) return bad("bad partition counts");
  return create_parts(init_n, lock_search, num_domains, parts_per_domain, 0, (uint64_t)num_domains * (uint64_t)parts_per_domain, devices,
                      n_devices, out);
}
int pppcsr_create_local(uint32_t init_n, int lock_search, int num_domains, int parts_per_domain, uint64_t first_part,
                        uint64_t n_local_parts, int device, pppcsr_t *out) {
  return create_parts(init_n, lock_search, num_domains, parts_per_domain, first_part, n_local_parts, &device, 1, out);
}
int pppcsr_destroy(pppcsr_t h) {
  if (!h) return 0;
  for (auto *q : h->parts) ppcsr_destroy(q);
  if (h->d_bucketed) capi_dev_free(h->d_bucketed);
  if (h->d_counts) capi_dev_free(h->d_counts);
  if (h->d_moved) capi_dev_free(h->d_moved);
  delete h;
  return 0;
}
#define PP_CHECK() \
  if (!h) return bad("null handle")
#define PP_PART(k) \
  if ((k) >= h->parts.size() || !h->parts[(k)]) return bad("partition not resident in this process")
int pppcsr_num_partitions(pppcsr_t h, uint64_t *out) { PP_CHECK(); *out = h->parts.size(); return 0; }
int pppcsr_get_partition(pppcsr_t h, uint64_t v, uint64_t *part) { PP_CHECK(); *part = owner_of(h->distribution, v); return 0; }
int pppcsr_partition_start(pppcsr_t h, uint64_t part, uint64_t *first) {
  PP_CHECK();
  if (part >= h->parts.size()) return bad("partition out of range");
  *first = h->distribution[part];
  return 0;
}
int pppcsr_partition(pppcsr_t h, uint64_t part, ppcsr_t *out) {
  PP_CHECK();
  if (part >= h->parts.size()) return bad("partition out of range");
  PP_PART(part);
  *out = h->parts[part];
  return 0;
}
int pppcsr_add_edge(pppcsr_t h, uint32_t s, uint32_t d, uint32_t v) {
  PP_CHECK();
  const uint64_t k = owner_of(h->distribution, s);
  PP_PART(k);
  return ppcsr_add_edge(h->parts[k], (uint32_t)(s - h->distribution[k]), d, v);
}
int pppcsr_remove_edge(pppcsr_t h, uint32_t s, uint32_t d) {
  PP_CHECK();
  const uint64_t k = owner_of(h->distribution, s);
  PP_PART(k);
  return ppcsr_remove_edge(h->parts[k], (uint32_t)(s - h->distribution[k]), d);
}
int pppcsr_edge_exists(pppcsr_t h, uint32_t s, uint32_t d, int *exists) {
  PP_CHECK();
  const uint64_t k = owner_of(h->distribution, s);
  PP_PART(k);
  return ppcsr_edge_exists(h->parts[k], (uint32_t)(s - h->distribution[k]), d, exists);
}
int pppcsr_get_neighbourhood(pppcsr_t h, int src, int *out, uint64_t cap, uint64_t *count) {
  PP_CHECK();
  if (src < 0) { if (count) *count = 0; return 0; }
  const uint64_t k = owner_of(h->distribution, (uint64_t)src);
  PP_PART(k);
  return ppcsr_get_neighbourhood(h->parts[k], (int)((uint64_t)src - h->distribution[k]), out, cap, count);
}
int pppcsr_get_node(pppcsr_t h, uint32_t v, ppcsr_node *out) {
  PP_CHECK();
  const uint64_t k = owner_of(h->distribution, v);
  PP_PART(k);
  return ppcsr_get_node(h->parts[k], (uint32_t)(v - h->distribution[k]), out);
}
int pppcsr_get_n(pppcsr_t h, uint64_t *n) {
  PP_CHECK();
  uint64_t t = 0;
  for (auto *q : h->parts) { uint64_t x = 0; if (q) ppcsr_get_n(q, &x); t += x; }  // (resident partitions)
  *n = t;
  return 0;
}
// PPPCSR.cpp:44: the new vertex joins the last partition.  Every rank calls this (the vertex count is part of the layout
// pppcsr_repartition works from); only the rank that holds the last partition has an engine to grow.
int pppcsr_add_node(pppcsr_t h) {
  PP_CHECK();
  h->total_n++;
  if (!h->parts.back()) return 0;
  return ppcsr_add_node(h->parts.back());
}

static int bucket_host(const std::vector<uint64_t> &dist, const ppcsr_op *ops, uint64_t n, ppcsr_op *bucketed, uint64_t *counts);
int pppcsr_bucket_ops(uint32_t init_n, uint64_t n_parts, const ppcsr_op *ops, uint64_t n, ppcsr_op *bucketed, uint64_t *counts) {
  if (n_parts < 1 || (!ops && n) || !bucketed || !counts) return bad("bad arguments");
  std::vector<uint64_t> dist, sizes;
  partition_layout(init_n, n_parts, &dist, &sizes);
  return bucket_host(dist, ops, n, bucketed, counts);
}
static int bucket_host(const std::vector<uint64_t> &dist, const ppcsr_op *ops, uint64_t n, ppcsr_op *bucketed, uint64_t *counts) {
  const uint64_t n_parts = dist.size();
  std::vector<uint64_t> off(n_parts + 1, 0);
  for (uint64_t k = 0; k < n_parts; k++) counts[k] = 0;
  std::vector<uint32_t> owner(n);
  for (uint64_t i = 0; i < n; i++) {
    owner[i] = (uint32_t)owner_of(dist, ops[i].src);
    counts[owner[i]]++;
  }
  for (uint64_t k = 0; k < n_parts; k++) off[k + 1] = off[k] + counts[k];
  std::vector<uint64_t> cur(off.begin(), off.end() - 1);
  for (uint64_t i = 0; i < n; i++) {
    ppcsr_op o = ops[i];
    o.src = (uint32_t)(o.src - dist[owner[i]]);
    bucketed[cur[owner[i]]++] = o;
  }
  return 0;
}

static int bucket_device(const std::vector<uint64_t> &dist, const ppcsr_op *d_ops, uint64_t n, ppcsr_op *d_bucketed, uint64_t *d_counts,
                         void *stream) {
  const uint64_t n_parts = dist.size();
  if (n_parts < 1 || n_parts > 64 || (!d_ops && n) || !d_bucketed || !d_counts) return bad("bad arguments");
  std::vector<uint32_t> starts(n_parts);
  for (uint64_t k = 0; k < n_parts; k++) starts[k] = (uint32_t)dist[k];
  std::string msg;
  int rc = ppcsr::bucket_ops_device(starts.data(), (uint32_t)n_parts, reinterpret_cast<const ppcsr::Op *>(d_ops), n,
                                    reinterpret_cast<ppcsr::Op *>(d_bucketed), reinterpret_cast<unsigned long long *>(d_counts), stream, &msg);
  if (rc != 0) g_last_error = msg;
  return rc;
}
int pppcsr_bucket_ops_device(uint32_t init_n, uint64_t n_parts, const ppcsr_op *d_ops, uint64_t n, ppcsr_op *d_bucketed,
                             uint64_t *d_counts, void *stream) {
  if (n_parts < 1 || n_parts > 64) return bad("bad arguments");
  std::vector<uint64_t> dist, sizes;
  partition_layout(init_n, n_parts, &dist, &sizes);
  return bucket_device(dist, d_ops, n, d_bucketed, d_counts, stream);
}

// Partitions are independent engines with their own streams (PPPCSR.h:54): host threads drive them side by side — the
// reference runs every domain's workers concurrently (thread_pool_pppcsr.cpp:121-156) — so the latency-bound round
// kernels of different partitions overlap on the GPU(s).  Each partition still applies its own subsequence in stream order.
// ops[i] / counts[i] belong to partition first + i; `device_resident` selects the entry point.
// what the routed records are: updates (host / HBM), num_neighbors records, or the adds an EMPTY partition is bulk-built from
enum { PARTS_HOST = 0, PARTS_DEVICE = 1, PARTS_SET_NN = 2, PARTS_BULK = 3 };
static int apply_parts(pppcsr_t h, uint64_t first, uint64_t np, const ppcsr_op *const *ops, const uint64_t *counts, int kind) {
  for (uint64_t i = 0; i < np; i++)
    if (counts[i] && (first + i >= h->parts.size() || !h->parts[first + i])) return bad("partition not resident in this process");
  auto one = [&](uint64_t i) -> int {
    if (!counts[i]) return 0;
    if (kind == PARTS_SET_NN)
      return ret(h->parts[first + i]->e, h->parts[first + i]->e->set_num_neighbors_device(reinterpret_cast<const ppcsr::Op *>(ops[i]), counts[i]));
    if (kind == PARTS_BULK)
      return ret(h->parts[first + i]->e, h->parts[first + i]->e->bulk_build_device(reinterpret_cast<const ppcsr::Op *>(ops[i]), counts[i], nullptr));
    return kind == PARTS_DEVICE ? ppcsr_apply_batch_device(h->parts[first + i], ops[i], counts[i])
                                : ppcsr_apply_batch(h->parts[first + i], ops[i], counts[i]);
  };
#if defined(PPCSR_SIM)
  for (uint64_t i = 0; i < np; i++) {  // (the CPU emulator is single-threaded)
    const int rc = one(i);
    if (rc != 0) return rc;
  }
  return 0;
#else
  uint64_t T = std::min<uint64_t>(np, 16);
  if (const char *e = getenv("PPCSR_PP_THREADS")) T = std::max<uint64_t>(1, std::min<uint64_t>(np, strtoull(e, nullptr, 10)));  // measurement hook
  if (T <= 1) {
    for (uint64_t i = 0; i < np; i++) {
      const int rc = one(i);
      if (rc != 0) return rc;
    }
    return 0;
  }
  // largest subsequences first: with skewed partitions (raw RMAT labels put 44 % of the edges in partition 0) the batch
  // takes as long as its largest partition, which must not start last
  std::vector<uint64_t> order(np);
  for (uint64_t i = 0; i < np; i++) order[i] = i;
  std::sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) { return counts[a] > counts[b]; });
  std::vector<int> rcs(T, 0);
  std::vector<std::string> msgs(T);
  std::vector<std::thread> th;
  for (uint64_t t = 0; t < T; t++)
    th.emplace_back([&, t]() {
      for (uint64_t j = t; j < np; j += T) {
        const int r = one(order[j]);
        if (r != 0) {
          rcs[t] = r;
          msgs[t] = ppcsr_last_error();  // (thread-local in the worker)
          return;
        }
      }
    });
  for (auto &x : th) x.join();
  for (uint64_t t = 0; t < T; t++)
    if (rcs[t] != 0) {
      g_last_error = msgs[t];
      return rcs[t];
    }
  return 0;
#endif
}

int pppcsr_apply_batch(pppcsr_t h, const ppcsr_op *ops, uint64_t n) {
  PP_CHECK();
  if (n == 0) return 0;
  const uint64_t P = h->parts.size();
  std::vector<ppcsr_op> b(n);
  std::vector<uint64_t> counts(P);
  if (!ops) return bad("null ops");
  int rc = bucket_host(h->distribution, ops, n, b.data(), counts.data());
  if (rc != 0) return rc;
  std::vector<const ppcsr_op *> ptrs(P);
  uint64_t off = 0;
  for (uint64_t k = 0; k < P; k++) {
    ptrs[k] = b.data() + off;
    off += counts[k];
  }
  return apply_parts(h, 0, P, ptrs.data(), counts.data(), PARTS_HOST);
}

int pppcsr_apply_parts_device(pppcsr_t h, uint64_t first_part, uint64_t n_parts, const ppcsr_op *const *d_ops, const uint64_t *counts) {
  PP_CHECK();
  if (!d_ops || !counts) return bad("null argument");
  return apply_parts(h, first_part, n_parts, d_ops, counts, PARTS_DEVICE);
}

static int route_device(pppcsr_t h, const ppcsr_op *d_ops, uint64_t n, int kind) {
  PP_CHECK();
  if (n == 0) return 0;
  const uint64_t P = h->parts.size();
  if (P > 64) return bad("pppcsr_apply_batch_device: more than 64 partitions");
  for (uint64_t k = 0; k < P; k++)
    if (!h->parts[k] || h->device[k] != h->device[0]) return bad("pppcsr_apply_batch_device: every partition must be resident on the device that holds the batch");
  int rc = capi_set_device(h->device[0]);
  if (rc != 0) return rc;
  if (h->bucketed_cap < n) {
    if (h->d_bucketed) capi_dev_free(h->d_bucketed);
    h->d_bucketed = nullptr;
    h->bucketed_cap = 0;
    if (capi_dev_alloc((void **)&h->d_bucketed, n * sizeof(ppcsr_op)) != 0) return bad("pppcsr_apply_batch_device: out of device memory");
    h->bucketed_cap = n;
  }
  if (!h->d_counts && capi_dev_alloc((void **)&h->d_counts, 64 * sizeof(uint64_t)) != 0) return bad("pppcsr_apply_batch_device: out of device memory");
  rc = bucket_device(h->distribution, d_ops, n, h->d_bucketed, h->d_counts, nullptr);
  if (rc != 0) return rc;
  std::vector<uint64_t> counts(P);
  if (capi_d2h_sync(counts.data(), h->d_counts, P * sizeof(uint64_t)) != 0) return bad("pppcsr_apply_batch_device: device-to-host copy failed");
  std::vector<const ppcsr_op *> ptrs(P);
  uint64_t off = 0;
  for (uint64_t k = 0; k < P; k++) {
    ptrs[k] = h->d_bucketed + off;
    off += counts[k];
  }
  return apply_parts(h, 0, P, ptrs.data(), counts.data(), kind);
}
int pppcsr_apply_batch_device(pppcsr_t h, const ppcsr_op *d_ops, uint64_t n) { return route_device(h, d_ops, n, PARTS_DEVICE); }
int pppcsr_set_num_neighbors_device(pppcsr_t h, const ppcsr_op *d_recs, uint64_t n) { return route_device(h, d_recs, n, PARTS_SET_NN); }
int pppcsr_bulk_build_device(pppcsr_t h, const ppcsr_op *d_adds, uint64_t n) { return route_device(h, d_adds, n, PARTS_BULK); }

// ---- owner exchange in three steps: pack -> transport -> apply (thread_pool_pppcsr.cpp:96-118 replaced across processes) ----
// The staging object below knows nothing about the carrier: pppcsr_exchange_apply moves the bytes with RCCL, the tests move
// them with gloo between two CPU-emulator processes — pack, layout and apply are the same code in both.
}  // extern "C"
struct pppcsr_xchg {
  pppcsr_engine *h = nullptr;
  uint64_t h_gen = 0;                    // pppcsr_engine::gen of h when this staging was made
  int nranks = 1, rank = 0, device = 0;
  uint64_t ppr = 0, first = 0;           // this rank's partitions: [first, first + ppr)
  ppcsr_op *d_send = nullptr;            // the block, stably bucketed by owner partition (= by peer: ranks hold ascending ranges)
  uint64_t send_cap = 0;
  uint64_t *d_counts = nullptr;          // [P] rows per partition of d_send (device)
  uint64_t *d_rcounts = nullptr;         // [nranks * ppr] rows each source rank holds for my partitions (device; RCCL carrier)
  ppcsr_op *d_out = nullptr;             // per-partition streams, partition-major, source-rank-minor
  uint64_t out_cap = 0;
  std::vector<uint64_t> send_counts, send_off, recv_counts, part_first, part_count;
  std::vector<ppcsr_op *> dst;           // [nranks * ppr] where the segment (source r, local partition q) lands: dst[r * ppr + q]
  int stage = 0;                         // 0 idle, 1 packed, 2 laid out
};
static void xchg_free(pppcsr_xchg *x) {
  if (!x) return;
  capi_set_device(x->device);
  for (void *q : {(void *)x->d_send, (void *)x->d_counts, (void *)x->d_rcounts, (void *)x->d_out})
    if (q) capi_dev_free(q);
  delete x;
}
// bucketing only, asynchronous on `stream` (the RCCL carrier ships d_counts without reading it back first)
static int xchg_pack_async(pppcsr_xchg *x, const ppcsr_op *d_ops, uint64_t n, void *stream) {
  if (!d_ops && n) return bad("null ops");
  int rc = capi_set_device(x->device);
  if (rc != 0) return rc;
  if (x->send_cap < std::max<uint64_t>(n, 1)) {
    if (x->d_send) capi_dev_free(x->d_send);
    x->d_send = nullptr;
    x->send_cap = 0;
    if (capi_dev_alloc((void **)&x->d_send, std::max<uint64_t>(n, 1) * sizeof(ppcsr_op)) != 0) return bad("out of device memory (exchange send block)");
    x->send_cap = std::max<uint64_t>(n, 1);
  }
  x->stage = 0;
  return bucket_device(x->h->distribution, d_ops, n, x->d_send, x->d_counts, stream);
}
static void xchg_set_send_counts(pppcsr_xchg *x) {
  const uint64_t P = x->h->parts.size();
  x->send_off.assign(P + 1, 0);
  for (uint64_t k = 0; k < P; k++) x->send_off[k + 1] = x->send_off[k] + x->send_counts[k];
  x->stage = 1;
}
extern "C" {
int pppcsr_xchg_create(pppcsr_t h, int n_ranks, int rank, pppcsr_xchg_t *out) {
  PP_CHECK();
  if (!out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return bad("bad rank arguments");
  *out = nullptr;
  const uint64_t P = h->parts.size();
  if (P > 64 || P % (uint64_t)n_ranks) return bad("partitions must be a multiple of the ranks (and at most 64)");
  const uint64_t ppr = P / (uint64_t)n_ranks, first = (uint64_t)rank * ppr;
  // this rank must hold exactly the contiguous range [rank * ppr, (rank + 1) * ppr) of the global layout, on one device
  for (uint64_t k = 0; k < P; k++) {
    const bool mine = k >= first && k < first + ppr;
    if (mine != (h->parts[k] != nullptr)) return bad("the resident partitions are not this rank's range [rank * P / ranks, (rank + 1) * P / ranks)");
    if (mine && h->device[k] != h->device[first]) return bad("a rank's partitions must live on one device");
  }
  std::unique_ptr<pppcsr_xchg> x(new pppcsr_xchg());
  x->h = h;
  x->h_gen = h->gen;
  x->nranks = n_ranks;
  x->rank = rank;
  x->ppr = ppr;
  x->first = first;
  x->device = h->device[first];
  int rc = capi_set_device(x->device);
  if (rc != 0) return rc;
  if (capi_dev_alloc((void **)&x->d_counts, 64 * sizeof(uint64_t)) != 0 || capi_dev_alloc((void **)&x->d_rcounts, 64 * sizeof(uint64_t)) != 0) {
    xchg_free(x.release());
    return bad("out of device memory");
  }
  x->send_counts.assign(P, 0);
  x->recv_counts.assign((uint64_t)n_ranks * ppr, 0);
  x->part_first.assign(ppr, 0);
  x->part_count.assign(ppr, 0);
  x->dst.assign((uint64_t)n_ranks * ppr, nullptr);
  *out = x.release();
  return 0;
}
int pppcsr_xchg_destroy(pppcsr_xchg_t x) {
  xchg_free(x);
  return 0;
}
int pppcsr_xchg_pack(pppcsr_xchg_t x, const ppcsr_op *d_ops, uint64_t n, uint64_t *send_counts, const ppcsr_op **d_send) {
  if (!x) return bad("null exchange");
  int rc = xchg_pack_async(x, d_ops, n, nullptr);
  if (rc != 0) return rc;
  const uint64_t P = x->h->parts.size();
  if (capi_d2h_sync(x->send_counts.data(), x->d_counts, P * sizeof(uint64_t)) != 0) return bad("pppcsr_xchg_pack: device-to-host copy failed");
  xchg_set_send_counts(x);
  if (send_counts) memcpy(send_counts, x->send_counts.data(), P * sizeof(uint64_t));
  if (d_send) *d_send = x->d_send;
  return 0;
}
int pppcsr_xchg_layout(pppcsr_xchg_t x, const uint64_t *recv_counts, ppcsr_op **d_dst) {
  if (!x || !recv_counts) return bad("null argument");
  if (x->stage < 1) return bad("pppcsr_xchg_layout before pppcsr_xchg_pack");
  const uint64_t W = (uint64_t)x->nranks, ppr = x->ppr;
  for (uint64_t q = 0; q < ppr; q++)  // what a rank sends to itself it knows already: a carrier that disagrees is broken
    if (recv_counts[(uint64_t)x->rank * ppr + q] != x->send_counts[x->first + q]) return bad("pppcsr_xchg_layout: own segment sizes do not match the packed block");
  uint64_t total = 0;
  for (uint64_t i = 0; i < W * ppr; i++) {
    x->recv_counts[i] = recv_counts[i];
    total += recv_counts[i];
  }
  int rc = capi_set_device(x->device);
  if (rc != 0) return rc;
  if (x->out_cap < std::max<uint64_t>(total, 1)) {
    if (x->d_out) capi_dev_free(x->d_out);
    x->d_out = nullptr;
    x->out_cap = 0;
    if (capi_dev_alloc((void **)&x->d_out, std::max<uint64_t>(total, 1) * sizeof(ppcsr_op)) != 0) return bad("out of device memory (exchange streams)");
    x->out_cap = std::max<uint64_t>(total, 1);
  }
  uint64_t run = 0;
  for (uint64_t q = 0; q < ppr; q++) {  // partition-major, source-minor: sources in rank order = global stream order
    x->part_first[q] = run;
    for (uint64_t r = 0; r < W; r++) {
      x->dst[r * ppr + q] = x->d_out + run;
      run += recv_counts[r * ppr + q];
    }
    x->part_count[q] = run - x->part_first[q];
  }
  if (d_dst) memcpy(d_dst, x->dst.data(), W * ppr * sizeof(ppcsr_op *));
  x->stage = 2;
  return 0;
}
static int xchg_finish(pppcsr_xchg_t x, int kind) {
  if (!x) return bad("null exchange");
  if (x->stage != 2) return bad("pppcsr_xchg_apply before pppcsr_xchg_layout");
  x->stage = 0;
  std::vector<const ppcsr_op *> ptrs(x->ppr);
  for (uint64_t q = 0; q < x->ppr; q++) ptrs[q] = x->d_out + x->part_first[q];
  return apply_parts(x->h, x->first, x->ppr, ptrs.data(), x->part_count.data(), kind);
}
int pppcsr_xchg_apply(pppcsr_xchg_t x) { return xchg_finish(x, PARTS_DEVICE); }
int pppcsr_xchg_set_num_neighbors(pppcsr_xchg_t x) { return xchg_finish(x, PARTS_SET_NN); }
int pppcsr_xchg_bulk_build(pppcsr_xchg_t x) { return xchg_finish(x, PARTS_BULK); }
}  // extern "C"

// ---- the RCCL carrier: grouped ncclSend / ncclRecv on a HIP stream, no torch in the data path ----
struct pppcsr_comm {
  ppcsr_xchg *t = nullptr;     // communicator + stream (engine.cc)
  pppcsr_xchg *x = nullptr;    // staging for the PPPCSR it was last used with
  uint64_t *d_flags = nullptr; // [3][64] device words: status out / status in (and counts out / in of a rank without staging)
};
extern "C" {
int pppcsr_comm_unique_id(void *id_out_128_bytes) {
  if (!id_out_128_bytes) return bad("null output");
  std::string msg;
  const int rc = capi_xchg_unique_id(id_out_128_bytes, &msg);
  if (rc != 0) g_last_error = msg;
  return rc;
}
int pppcsr_comm_create(const void *id_128_bytes, int n_ranks, int rank, int device, pppcsr_comm_t *out) {
  if (!id_128_bytes || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return bad("bad communicator arguments");
  *out = nullptr;
  std::string msg;
  ppcsr_xchg *t = nullptr;
  const int rc = capi_xchg_create(id_128_bytes, n_ranks, rank, device, &t, &msg);
  if (rc != 0) {
    g_last_error = msg;
    return rc;
  }
  pppcsr_comm *c = new pppcsr_comm();
  c->t = t;
  *out = c;
  return 0;
}
int pppcsr_comm_destroy(pppcsr_comm_t c) {
  if (!c) return 0;
  xchg_free(c->x);
  if (c->d_flags) capi_dev_free(c->d_flags);
  capi_xchg_destroy(c->t);
  delete c;
  return 0;
}
static int exchange_run(pppcsr_t h, pppcsr_comm_t c, const ppcsr_op *d_ops, uint64_t n, int kind) {
  PP_CHECK();
  if (!c) return bad("null communicator");
  int W = 0, rank = 0, dev = 0;
  void *stream = nullptr;
  if (capi_xchg_ranks(c->t, &W, &rank, &dev, &stream) != 0) return bad("bad communicator");
  // Every rank takes part in every collective step of this call whatever happened locally: a rank that returned between two
  // steps would leave its peers blocked in their grouped ncclSend / ncclRecv for ever.  Local failures are carried along
  // (`failed`) and travel inside the steps themselves: (1) poisoned counts — all ranks skip the rows; (2) a status word after
  // the receive buffers have been laid out — again all ranks skip the rows.  Only a failure of the carrier itself returns at
  // once (nothing collective can follow it).
  int failed = 0;
  std::string first_msg;
  // (the staging is keyed on the handle's generation, not its address: a destroyed PPPCSR's address can come back)
  if (!c->x || c->x->h != h || c->x->h_gen != h->gen || c->x->nranks != W || c->x->rank != rank) {
    xchg_free(c->x);
    c->x = nullptr;
    failed = pppcsr_xchg_create(h, W, rank, &c->x);  // (the layout checks against the communicator's ranks happen here)
    if (failed != 0) first_msg = g_last_error;
  }
  if (failed == 0 && c->x->device != dev) {
    failed = bad("the communicator's device does not hold this rank's partitions");
    first_msg = g_last_error;
  }
  const uint64_t P = h->parts.size();
  if (P == 0 || P > 64 || P % (uint64_t)W) return bad("partitions must be a multiple of the ranks (and at most 64)");  // (the same on every rank)
  const uint64_t ppr = P / (uint64_t)W, nseg = (uint64_t)W * ppr;
  if (!c->d_flags && (capi_set_device(dev) != 0 || capi_dev_alloc((void **)&c->d_flags, 3 * 64 * sizeof(uint64_t)) != 0))
    return bad("pppcsr_exchange_apply: out of device memory (status words)");  // (first call only, before anything collective)
  pppcsr_xchg *x = c->x;  // may be null (failed): the counts then travel from / to the communicator's own words
  uint64_t *d_counts = x ? x->d_counts : c->d_flags + 64, *d_rcounts = x ? x->d_rcounts : c->d_flags + 128;
  if (failed == 0) {
    failed = xchg_pack_async(x, d_ops, n, stream);
    if (failed != 0) first_msg = g_last_error;
  }
  if (failed != 0 && capi_dev_memset(d_counts, 0xFF, 64 * sizeof(uint64_t), stream) != 0) return bad("pppcsr_exchange_apply: device memset failed");
  std::vector<const void *> sp(nseg);
  std::vector<void *> rp(nseg);
  std::vector<uint64_t> sb(nseg), rb(nseg);
  std::vector<int> peer(nseg);
  // (1) the counts: ppr numbers to and from every peer, straight from the bucketing kernels' output
  for (int r = 0; r < W; r++) {
    sp[r] = d_counts + (uint64_t)r * ppr;
    rp[r] = d_rcounts + (uint64_t)r * ppr;
    sb[r] = rb[r] = ppr * sizeof(uint64_t);
    peer[r] = r;
  }
  int rc = capi_xchg_sendrecv(c->t, (uint64_t)W, sp.data(), sb.data(), peer.data(), rp.data(), rb.data(), peer.data());
  if (rc != 0) {
    g_last_error = capi_xchg_error(c->t);
    return rc;  // the carrier itself is broken: nothing collective can follow
  }
  std::vector<uint64_t> rcounts(nseg, 0), scounts(P, 0);
  if (capi_d2h_sync(scounts.data(), d_counts, P * sizeof(uint64_t)) != 0 || capi_d2h_sync(rcounts.data(), d_rcounts, nseg * sizeof(uint64_t)) != 0)
    return bad("pppcsr_exchange_apply: device-to-host copy failed");  // (the runtime itself is broken)
  bool peer_failed = false;
  for (uint64_t i = 0; i < nseg; i++) peer_failed |= rcounts[i] == ~0ull;
  if (failed != 0 || peer_failed) {  // every rank sees a poisoned row of counts (a failed rank poisons all of its own): all leave here
    if (failed != 0) {
      g_last_error = first_msg;
      return failed;
    }
    return bad("pppcsr_exchange_apply: a peer rank failed to bucket its block; no rows were exchanged");
  }
  x->send_counts = scounts;
  xchg_set_send_counts(x);
  // (2) lay out the receive buffers, then agree that every rank could: one status word to and from every peer
  failed = pppcsr_xchg_layout(x, rcounts.data(), nullptr);  // (out of device memory for the received streams)
  if (failed != 0) first_msg = g_last_error;
  {
    const uint64_t word = failed != 0 ? 1ull : 0ull;
    std::vector<uint64_t> mine((size_t)W, word), theirs((size_t)W, 0);
    if (capi_h2d_sync(c->d_flags, mine.data(), (size_t)W * sizeof(uint64_t)) != 0) return bad("pppcsr_exchange_apply: host-to-device copy failed");
    for (int r = 0; r < W; r++) {
      sp[r] = c->d_flags + r;
      rp[r] = c->d_flags + 64 + r;
      sb[r] = rb[r] = sizeof(uint64_t);
      peer[r] = r;
    }
    rc = capi_xchg_sendrecv(c->t, (uint64_t)W, sp.data(), sb.data(), peer.data(), rp.data(), rb.data(), peer.data());
    if (rc != 0) {
      g_last_error = capi_xchg_error(c->t);
      return rc;
    }
    if (capi_d2h_sync(theirs.data(), c->d_flags + 64, (size_t)W * sizeof(uint64_t)) != 0) return bad("pppcsr_exchange_apply: device-to-host copy failed");
    bool any = false;
    for (int r = 0; r < W; r++) any |= theirs[r] != 0;
    if (failed != 0) {
      x->stage = 0;
      g_last_error = first_msg;
      return failed;
    }
    if (any) {
      x->stage = 0;
      return bad("pppcsr_exchange_apply: a peer rank could not lay out its receive buffers; no rows were exchanged");
    }
  }
  // (3) the rows: segment (peer r, partition q) leaves from the bucketed block and lands where the stream of q wants it
  for (int r = 0; r < W; r++)
    for (uint64_t q = 0; q < ppr; q++) {
      const uint64_t i = (uint64_t)r * ppr + q;
      sp[i] = x->d_send + x->send_off[i];
      sb[i] = x->send_counts[i] * sizeof(ppcsr_op);
      rp[i] = x->dst[i];
      rb[i] = rcounts[i] * sizeof(ppcsr_op);
      peer[i] = r;
    }
  rc = capi_xchg_sendrecv(c->t, nseg, sp.data(), sb.data(), peer.data(), rp.data(), rb.data(), peer.data());
  if (rc != 0) {
    g_last_error = capi_xchg_error(c->t);
    return rc;
  }
  return xchg_finish(x, kind);
}
int pppcsr_exchange_apply(pppcsr_t h, pppcsr_comm_t c, const ppcsr_op *d_ops, uint64_t n) { return exchange_run(h, c, d_ops, n, PARTS_DEVICE); }
int pppcsr_exchange_set_num_neighbors(pppcsr_t h, pppcsr_comm_t c, const ppcsr_op *d_recs, uint64_t n) {
  return exchange_run(h, c, d_recs, n, PARTS_SET_NN);
}
int pppcsr_exchange_bulk_build(pppcsr_t h, pppcsr_comm_t c, const ppcsr_op *d_adds, uint64_t n) { return exchange_run(h, c, d_adds, n, PARTS_BULK); }
}  // extern "C"

extern "C" {
// ---- repartitioning (the reference only sketches it: PCSR.h:91-112 was never implemented) ----
// The vertex ranges move; the edges of every partition whose range changes leave it as adds of the global stream
// (src global, array order = ascending (src, dest)), the partition is recreated empty at its new size, and the caller
// routes the adds with the ordinary machinery (owner bucketing by the new starts, the exchange across ranks) into the BULK
// BUILD of the recreated partitions (pppcsr_bulk_build_device / pppcsr_exchange_bulk_build / pppcsr_xchg_bulk_build): a
// source-sorted stream of a whole partition through the exact one-by-one path is the hot-vertex worst case (minutes for
// 10^8 edges), and there is no reference layout to reproduce anyway.  Partitions whose range stays keep their array
// untouched.  Scheduler options set on a recreated partition's engine return to their defaults.
int pppcsr_repartition_export(pppcsr_t h, const uint64_t *new_starts, const ppcsr_op **d_ops, uint64_t *n_out, const ppcsr_op **d_nn,
                              uint64_t *n_nn) {
  PP_CHECK();
  if (!new_starts || !d_ops || !n_out || !d_nn || !n_nn) return bad("null argument");
  *d_ops = *d_nn = nullptr;
  *n_out = *n_nn = 0;
  const uint64_t P = h->parts.size();
  if (new_starts[0] != 0) return bad("pppcsr_repartition: the first partition starts at vertex 0");
  for (uint64_t k = 1; k < P; k++)
    if (new_starts[k] < new_starts[k - 1]) return bad("pppcsr_repartition: starts must not decrease");
  if (new_starts[P - 1] > h->total_n) return bad("pppcsr_repartition: a partition starts past the last vertex");
  int dev = -1;
  for (uint64_t k = 0; k < P; k++)
    if (h->parts[k]) {
      if (dev < 0) dev = h->device[k];
      if (h->device[k] != dev) return bad("pppcsr_repartition: this process's partitions must live on one device");
    }
  auto end_of = [&](const uint64_t *st, uint64_t k) { return k + 1 < P ? st[k + 1] : h->total_n; };
  std::vector<uint64_t> cnt(P, 0), nk(P, 0);
  std::vector<char> changed(P, 0);
  uint64_t total = 0, total_nn = 0;
  for (uint64_t k = 0; k < P; k++) {
    if (!h->parts[k]) continue;
    changed[k] = h->distribution[k] != new_starts[k] || end_of(h->distribution.data(), k) != end_of(new_starts, k);
    if (!changed[k]) continue;
    int rc = ret(h->parts[k]->e, h->parts[k]->e->export_triples_device(0, nullptr, 0, &cnt[k]));
    if (rc != 0) return rc;
    total += cnt[k];
    ppcsr_get_n(h->parts[k], &nk[k]);
    total_nn += nk[k];
  }
  if (dev >= 0) {
    int rc = capi_set_device(dev);
    if (rc != 0) return rc;
  }
  if (total + total_nn > h->moved_cap) {  // [edges | num_neighbors records]
    if (h->d_moved) capi_dev_free(h->d_moved);
    h->d_moved = nullptr;
    h->moved_cap = 0;
    if (capi_dev_alloc((void **)&h->d_moved, (total + total_nn) * sizeof(ppcsr_op)) != 0) return bad("pppcsr_repartition: out of device memory");
    h->moved_cap = total + total_nn;
  }
  uint64_t off = 0, noff = total;
  for (uint64_t k = 0; k < P; k++) {
    if (!changed[k]) continue;
    uint64_t t = 0;
    int rc = ret(h->parts[k]->e, h->parts[k]->e->export_triples_device((uint32_t)h->distribution[k], reinterpret_cast<ppcsr::Op *>(h->d_moved) + off,
                                                                       cnt[k], &t));
    if (rc != 0) return rc;
    off += cnt[k];
    rc = ret(h->parts[k]->e, h->parts[k]->e->export_num_neighbors_device((uint32_t)h->distribution[k], reinterpret_cast<ppcsr::Op *>(h->d_moved) + noff));
    if (rc != 0) return rc;
    noff += nk[k];
  }
  // All or nothing: every replacement engine is created BEFORE any old partition goes.  A failure (out of device memory for
  // partition k) destroys the replacements made so far and leaves the handle as it was — every partition, its edges and the
  // layout untouched; only the exported copy in d_moved has been written.
  std::vector<ppcsr_t> fresh(P, nullptr);
  for (uint64_t k = 0; k < P; k++) {
    if (!changed[k]) continue;
    const uint64_t size = end_of(new_starts, k) - new_starts[k];
    int rc = ppcsr_create((uint32_t)size, (uint32_t)size, h->lock_search, h->device[k], &fresh[k]);
    if (rc != 0) {
      const std::string why = g_last_error;
      for (uint64_t q = 0; q < k; q++)
        if (fresh[q]) ppcsr_destroy(fresh[q]);
      g_last_error = why;
      return rc;
    }
  }
  for (uint64_t k = 0; k < P; k++) {
    if (!changed[k]) continue;
    ppcsr_destroy(h->parts[k]);
    h->parts[k] = fresh[k];
  }
  for (uint64_t k = 0; k < P; k++) h->distribution[k] = new_starts[k];
  *d_ops = h->d_moved;
  *n_out = total;
  *d_nn = h->d_moved + total;
  *n_nn = total_nn;
  return 0;
}
int pppcsr_repartition(pppcsr_t h, const uint64_t *new_starts) {
  PP_CHECK();
  for (auto *q : h->parts)
    if (!q) return bad("pppcsr_repartition: not every partition is resident here (use pppcsr_repartition_export + pppcsr_exchange_apply)");
  const ppcsr_op *d = nullptr, *dn = nullptr;
  uint64_t n = 0, nn = 0;
  int rc = pppcsr_repartition_export(h, new_starts, &d, &n, &dn, &nn);
  if (rc != 0) return rc;
  rc = pppcsr_bulk_build_device(h, d, n);
  if (rc != 0) return rc;
  return pppcsr_set_num_neighbors_device(h, dn, nn);
}
// starts of P contiguous vertex ranges of about equal weight, weight(v) = num_neighbors(v) + 1 (edges are what updates
// cost; the + 1 keeps empty stretches from collapsing into one partition).  All partitions resident.
int pppcsr_balanced_starts(pppcsr_t h, uint64_t *starts_out) {
  PP_CHECK();
  if (!starts_out) return bad("null output");
  const uint64_t P = h->parts.size();
  std::vector<uint64_t> w;
  w.reserve(h->total_n);
  for (uint64_t k = 0; k < P; k++) {
    if (!h->parts[k]) return bad("pppcsr_balanced_starts: not every partition is resident here");
    uint64_t nk = 0;
    ppcsr_get_n(h->parts[k], &nk);
    std::vector<ppcsr_node> nodes(nk);
    if (nk) {
      int rc = ppcsr_export_state(h->parts[k], nullptr, nodes.data());
      if (rc != 0) return rc;
    }
    for (uint64_t v = 0; v < nk; v++) {
      const int32_t nn = (int32_t)nodes[v].num_neighbors;  // (a delete of a missing edge can leave it below zero: PCSR.cpp:1409)
      w.push_back((uint64_t)(nn > 0 ? nn : 0) + 1);
    }
  }
  uint64_t total = 0;
  for (uint64_t x : w) total += x;
  uint64_t run = 0, v = 0;
  starts_out[0] = 0;
  for (uint64_t k = 1; k < P; k++) {
    const uint64_t goal = (total * k) / P;  // cumulative weight the first k partitions should reach
    while (v < w.size() && run + w[v] / 2 < goal) run += w[v++];
    starts_out[k] = v;
  }
  return 0;
}
}  // extern "C"
