// C ABI (include/ppcsr.h) over ppcsr::Engine.  No torch types, plain pointers and sizes.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
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
#if !defined(PPCSR_SIM)
struct ppcsr_xchg;
int capi_xchg_unique_id(void *out128, std::string *err);
int capi_xchg_create(const void *id128, int nranks, int rank, int device, ppcsr_xchg **out, std::string *err);
int capi_xchg_destroy(ppcsr_xchg *x);
int capi_xchg_route(ppcsr_xchg *x, uint32_t init_n, uint32_t n_parts, const ppcsr::Op *d_ops, uint64_t n, uint64_t cap, const ppcsr::Op **out_ptrs,
                    uint64_t *out_counts);
const char *capi_xchg_error(ppcsr_xchg *x);
#endif

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
  out->chained = s.chained;
  return 0;
}
int ppcsr_set_option(ppcsr_t h, const char *key, int64_t value) { H_CHECK(); return ret(h->e, h->e->set_option(key, value)); }
int ppcsr_check_invariants(ppcsr_t h, uint64_t *bad_leaves) { H_CHECK(); return ret(h->e, h->e->check_invariants(bad_leaves)); }
int ppcsr_bench_scan_all(ppcsr_t h, double *ms, uint64_t *total) { H_CHECK(); return ret(h->e, h->e->scan_all_device(ms, total)); }
int ppcsr_bench_rebalance(ppcsr_t h, uint64_t w, int iters, double *ms) { H_CHECK(); return ret(h->e, h->e->rebalance_bench(w, iters, ms)); }
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
  // device-resident routing scratch of pppcsr_apply_batch_device (bucketed copy of the batch + bucket sizes)
  ppcsr_op *d_bucketed = nullptr;
  uint64_t bucketed_cap = 0;
  uint64_t *d_counts = nullptr;
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
int pppcsr_add_node(pppcsr_t h) { PP_CHECK(); PP_PART(h->parts.size() - 1); return ppcsr_add_node(h->parts.back()); }  // PPPCSR.cpp:44

int pppcsr_bucket_ops(uint32_t init_n, uint64_t n_parts, const ppcsr_op *ops, uint64_t n, ppcsr_op *bucketed, uint64_t *counts) {
  if (n_parts < 1 || (!ops && n) || !bucketed || !counts) return bad("bad arguments");
  std::vector<uint64_t> dist, sizes;
  partition_layout(init_n, n_parts, &dist, &sizes);
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

int pppcsr_bucket_ops_device(uint32_t init_n, uint64_t n_parts, const ppcsr_op *d_ops, uint64_t n, ppcsr_op *d_bucketed,
                             uint64_t *d_counts, void *stream) {
  if (n_parts < 1 || n_parts > 64 || (!d_ops && n) || !d_bucketed || !d_counts) return bad("bad arguments");
  std::string msg;
  int rc = ppcsr::bucket_ops_device(init_n, (uint32_t)n_parts, reinterpret_cast<const ppcsr::Op *>(d_ops), n,
                                    reinterpret_cast<ppcsr::Op *>(d_bucketed), reinterpret_cast<unsigned long long *>(d_counts), stream, &msg);
  if (rc != 0) g_last_error = msg;
  return rc;
}

// Partitions are independent engines with their own streams (PPPCSR.h:54): host threads drive them side by side — the
// reference runs every domain's workers concurrently (thread_pool_pppcsr.cpp:121-156) — so the latency-bound round
// kernels of different partitions overlap on the GPU(s).  Each partition still applies its own subsequence in stream order.
// ops[i] / counts[i] belong to partition first + i; `device_resident` selects the entry point.
static int apply_parts(pppcsr_t h, uint64_t first, uint64_t np, const ppcsr_op *const *ops, const uint64_t *counts, bool device_resident) {
  for (uint64_t i = 0; i < np; i++)
    if (counts[i] && (first + i >= h->parts.size() || !h->parts[first + i])) return bad("partition not resident in this process");
  auto one = [&](uint64_t i) -> int {
    if (!counts[i]) return 0;
    return device_resident ? ppcsr_apply_batch_device(h->parts[first + i], ops[i], counts[i])
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
  int rc = pppcsr_bucket_ops(h->init_n, P, ops, n, b.data(), counts.data());
  if (rc != 0) return rc;
  std::vector<const ppcsr_op *> ptrs(P);
  uint64_t off = 0;
  for (uint64_t k = 0; k < P; k++) {
    ptrs[k] = b.data() + off;
    off += counts[k];
  }
  return apply_parts(h, 0, P, ptrs.data(), counts.data(), false);
}

int pppcsr_apply_parts_device(pppcsr_t h, uint64_t first_part, uint64_t n_parts, const ppcsr_op *const *d_ops, const uint64_t *counts) {
  PP_CHECK();
  if (!d_ops || !counts) return bad("null argument");
  return apply_parts(h, first_part, n_parts, d_ops, counts, true);
}

int pppcsr_apply_batch_device(pppcsr_t h, const ppcsr_op *d_ops, uint64_t n) {
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
  rc = pppcsr_bucket_ops_device(h->init_n, P, d_ops, n, h->d_bucketed, h->d_counts, nullptr);
  if (rc != 0) return rc;
  std::vector<uint64_t> counts(P);
  if (capi_d2h_sync(counts.data(), h->d_counts, P * sizeof(uint64_t)) != 0) return bad("pppcsr_apply_batch_device: device-to-host copy failed");
  std::vector<const ppcsr_op *> ptrs(P);
  uint64_t off = 0;
  for (uint64_t k = 0; k < P; k++) {
    ptrs[k] = h->d_bucketed + off;
    off += counts[k];
  }
  return apply_parts(h, 0, P, ptrs.data(), counts.data(), true);
}

// ---- native exchange: RCCL send/recv on a HIP stream, no torch in the data path (thread_pool_pppcsr.cpp:96-118 replaced) ----
#if !defined(PPCSR_SIM)
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
  ppcsr_xchg *x = nullptr;
  const int rc = capi_xchg_create(id_128_bytes, n_ranks, rank, device, &x, &msg);
  if (rc != 0) {
    g_last_error = msg;
    return rc;
  }
  *out = reinterpret_cast<pppcsr_comm_t>(x);
  return 0;
}
int pppcsr_comm_destroy(pppcsr_comm_t c) { return capi_xchg_destroy(reinterpret_cast<ppcsr_xchg *>(c)); }
int pppcsr_exchange_apply(pppcsr_t h, pppcsr_comm_t c, const ppcsr_op *d_ops, uint64_t n, uint64_t capacity) {
  PP_CHECK();
  if (!c) return bad("null communicator");
  ppcsr_xchg *x = reinterpret_cast<ppcsr_xchg *>(c);
  const uint64_t P = h->parts.size();
  uint64_t first = P, nlocal = 0;
  for (uint64_t k = 0; k < P; k++)
    if (h->parts[k]) {
      if (first == P) first = k;
      nlocal++;
    }
  if (nlocal == 0 || nlocal > 64) return bad("no resident partitions");
  std::vector<const ppcsr::Op *> ptrs(nlocal);
  std::vector<uint64_t> counts(nlocal);
  int rc = capi_xchg_route(x, h->init_n, (uint32_t)P, reinterpret_cast<const ppcsr::Op *>(d_ops), n, capacity, ptrs.data(), counts.data());
  if (rc != 0) {
    g_last_error = capi_xchg_error(x);
    return rc;
  }
  return apply_parts(h, first, nlocal, reinterpret_cast<const ppcsr_op *const *>(ptrs.data()), counts.data(), true);
}
#else
int pppcsr_comm_unique_id(void *) { return bad("no RCCL in the CPU emulator build"); }
int pppcsr_comm_create(const void *, int, int, int, pppcsr_comm_t *) { return bad("no RCCL in the CPU emulator build"); }
int pppcsr_comm_destroy(pppcsr_comm_t) { return 0; }
int pppcsr_exchange_apply(pppcsr_t, pppcsr_comm_t, const ppcsr_op *, uint64_t, uint64_t) { return bad("no RCCL in the CPU emulator build"); }
#endif

}  // extern "C"
