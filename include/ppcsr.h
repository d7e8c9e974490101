/*
 * ppcsr.h — C ABI of the MI355X-native packed-CSR update engine (libppcsr_hip.so).
 *
 * This is the drop-in boundary for the PMA insert/delete + rebalance hot path of
 * domargan/parallel-packed-csr.  The reference has no FFI layer: its boundary is the C++ classes
 * PCSR (src/pcsr/PCSR.h:64-202) and PPPCSR (src/pppcsr/PPPCSR.h:11-60) consumed by the thread
 * pools (src/thread_pool/thread_pool.cpp:44-48, src/thread_pool_pppcsr/thread_pool_pppcsr.cpp:76-82).
 * Every entry point below names the reference member it replaces.  All functions return 0 on success
 * or a ppcsr_status code; none of them exits the process (the reference calls exit() on allocation
 * failure, PCSR.cpp:49-54).  There is NO CPU fallback: without a visible MI355X ppcsr_create fails.
 *
 * Semantics: a batch is applied with the result of the reference driven in stream order on one thread
 * (the only deterministic mode of the reference): identical N/logN/H, identical
 * edges[] bytes, identical nodes[] triples.
 *
 * Preconditions inherited from the reference's sentinel encoding (PCSR.cpp:64): dst != 0xFFFFFFFF and
 * edge value != 0xFFFFFFFF for user edges.
 */
#ifndef PPCSR_H
#define PPCSR_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ppcsr_engine *ppcsr_t;   /* one PCSR instance resident on one GPU            */
typedef struct pppcsr_engine *pppcsr_t; /* vertex-range partitioned set of PCSRs (PPPCSR)   */
typedef struct pppcsr_comm *pppcsr_comm_t; /* RCCL communicator + stream + staging of the native exchange */

/* reference edge_t (PCSR.h:30-35) and node_t (PCSR.h:18-23): same field order, same 12-byte layout */
typedef struct { uint32_t src, dest, value; } ppcsr_edge;
typedef struct { uint32_t beginning, end, num_neighbors; } ppcsr_node;
/* one update of a stream: op == 0 deletes (src,dst); op != 0 adds (src,dst) with edge value = op
 * (reference task record src/utility/task.h:10-15; the pools always add with value 1) */
typedef struct { uint32_t src, dst, op; } ppcsr_op;

typedef enum {
  PPCSR_STATUS_OK = 0,
  PPCSR_STATUS_EINVAL = 1,
  PPCSR_STATUS_ENOMEM = 2,
  PPCSR_STATUS_EHIP = 3,         /* HIP runtime error or no GPU */
  PPCSR_STATUS_EUNSUPPORTED = 4, /* no null slot on either side of a slide (PCSR.cpp:378-383), > 2^31 slots */
  PPCSR_STATUS_EINTERNAL = 5,
  PPCSR_STATUS_ERANGE = 6        /* output buffer too small; *count holds the needed size */
} ppcsr_status;

typedef struct {
  uint64_t N, n;
  int32_t logN, H;
  uint64_t rounds, committed, planned, exclusive_ops, round_syncs;
  uint64_t redistribute_calls, redistribute_slots; /* what the reference's redistribute() would move */
  uint64_t double_calls, half_calls, big_redistributes, rollbacks;
  uint64_t not_found, duplicates, noops, slide_slots;
  uint64_t ops_applied;
  double last_batch_ms;     /* device-only time of the last batch (ops already in HBM)  */
  double last_batch_h2d_ms; /* H2D time of the op array when it came from a host buffer */
  /* option "profile"=1: HIP-event time spent in each round kernel, measured on the engine's own stream */
  double prof_plan_ms, prof_check_ms, prof_apply_ms, prof_compact_ms;
  uint64_t prof_launches; /* launches of each of the three round kernels */
  uint64_t wasted_rounds; /* rounds of speculative epochs that were rolled back (not counted in `rounds`) */
  uint64_t narrow_lost;   /* how often that add_node path was taken */
  uint64_t narrow;        /* 1: regular structure (parallel rounds); 0: add_node after a doubling has left overlapping vertex ranges
                             (PCSR.cpp:533-540, 681-703): updates run one per round until a re-check finds the ranges sane again */
} ppcsr_stats_t;

/* PCSR::PCSR(init_n, src_n, lock_search, domain)  — PCSR.cpp:775-838; `device` replaces the NUMA domain */
int ppcsr_create(uint32_t init_n, uint32_t src_n, int lock_search, int device, ppcsr_t *out);
/* PCSR::~PCSR — PCSR.cpp:840-851 */
int ppcsr_destroy(ppcsr_t h);
/* PCSR::add_edge — PCSR.cpp:706, 1374-1445 */
int ppcsr_add_edge(ppcsr_t h, uint32_t src, uint32_t dst, uint32_t value);
/* PCSR::remove_edge — PCSR.cpp:709-773 */
int ppcsr_remove_edge(ppcsr_t h, uint32_t src, uint32_t dst);
/* PCSR::add_node — PCSR.cpp:681-703 */
int ppcsr_add_node(ppcsr_t h);
/* the worker loop of ThreadPool::execute (thread_pool.cpp:28-61): everything submitted before start() is
 * one batch; applied with sequential stream-order semantics.  Host buffer variant copies H2D first. */
int ppcsr_apply_batch(ppcsr_t h, const ppcsr_op *ops, uint64_t n);
/* same, ops already resident in this GPU's HBM (e.g. the output of the all-to-all exchange) */
int ppcsr_apply_batch_device(ppcsr_t h, const ppcsr_op *d_ops, uint64_t n);
/* PCSR::edge_exists — PCSR.cpp:860-869 */
int ppcsr_edge_exists(ppcsr_t h, uint32_t src, uint32_t dst, int *exists);
/* PCSR::get_n — PCSR.cpp:100 */
int ppcsr_get_n(ppcsr_t h, uint64_t *n);
/* PCSR::getNode — PCSR.h:118-124 */
int ppcsr_get_node(ppcsr_t h, uint32_t v, ppcsr_node *out);
/* edges.N / edges.logN / edges.H — PCSR.h:37-44 */
int ppcsr_geometry(ppcsr_t h, uint64_t *N, int *logN, int *H);
/* PCSR::get_neighbourhood — PCSR.cpp:901-912 (out may be NULL to query the count) */
int ppcsr_get_neighbourhood(ppcsr_t h, int src, int *out, uint64_t cap, uint64_t *count);
/* PCSR::read_neighbourhood — PCSR.cpp:892-899 */
int ppcsr_read_neighbourhood(ppcsr_t h, int src);
/* bulk neighbour scan: get_neighbourhood for every vertex at once as CSR (row_offsets[n+1], dests[total]) */
int ppcsr_scan_all(ppcsr_t h, uint64_t *row_offsets, int *dests, uint64_t cap, uint64_t *total);
/* Bulk build of an EMPTY graph from a list of adds — an explicit NON-parity fast path: the reference can
 * only build a graph by single inserts (src/main.cpp:160-189 feeds add_edge one line at a time) and the array layout that
 * produces depends on the insertion history.  This call yields a valid packed-memory array with the same neighbourhoods,
 * the same num_neighbors (every add counts, duplicates keep the last value) and the same invariants, but NOT the slot
 * layout of the one-by-one build; updates applied afterwards go through the ordinary (sequentially exact) path.  Entries
 * with op == 0 or src >= n are ignored, as add_edge ignores them.  Fails with EINVAL if the graph already holds edges. */
int ppcsr_bulk_build(ppcsr_t h, const ppcsr_op *adds, uint64_t n, double *device_ms);
/* Graph-algorithm consumers run on the device over the gapped array.
 * bfs — src/utility/bfs.h:15-36: levels[v] = BFS level of v from `start`, UINT32_MAX when unreachable (levels: n entries).
 * pagerank — src/utility/pagerank.h:15-29: one push step, out[d] = sum over edges (s, d), in ascending s, of
 *   node_values[s] / num_neighbors(s) in fp32 — added in the reference's order, so the result is bit-identical to the
 *   reference template's (node_values, out: n entries).  device_ms (may be NULL): device time of the traversal. */
int ppcsr_bfs(ppcsr_t h, uint32_t start, uint32_t *levels, double *device_ms);
int ppcsr_pagerank(ppcsr_t h, const float *node_values, float *out, double *device_ms);
/* raw state for parity checks: items[N], nodes[n] exactly as the reference holds them (PCSR.h:67,128) */
int ppcsr_export_state(ppcsr_t h, ppcsr_edge *items, ppcsr_node *nodes);
int ppcsr_stats(ppcsr_t h, ppcsr_stats_t *out);
/* knobs (int64 values; every key engine.cc's set_option accepts):
 *   scheduler  "mode" (0 strict prefix rounds, 1 speculative rounds + validated rollback; default 1), "opt_horizon" (round
 *              width cap; default 3 x "resident_waves" = 3 x CUs x 24), "start_horizon", "adaptive", "epoch_ops", "epoch_short", "epoch_grow_after",
 *              "epoch_adapt", "region_slots" / "region_wide" / "region_calm" / "region_rare" / "region_rare_calm" / "region_rare_dist" / "region_rare_cpr", "soft_barrier", "defer_barrier", "small_batch" (batches up to this size take the strict
 *              rounds), "max_horizon" / "min_horizon" / "init_horizon" (strict rounds), "rounds_per_sync"
 *   windows    "big_min" / "big_window" (slots: above big_min a workgroup of the round rebalances the window, above
 *              big_window the update is exclusive), "big_grid", "excl_in_wave"
 *   rebalance  "scatter_variant" (0 LDS-staged, 1 register runs, 2 runs + in-tile leaf scan), "scatter_blocks",
 *              "rb_tile", "rb_min_tiles", "rb_prefetch", "rb_inplace_min" (partial windows of at least this many slots are
 *              rebalanced in place; 0 = always through the scratch array), "rb_inplace_cpw", "rb_inplace_lists"
 *   search     "search_narrow" (0: literal binary walk only)
 *   measuring  "profile" (1: HIP events around every round kernel, reported through ppcsr_stats), "diag" (1: why updates
 *              did not commit, per epoch, on stderr; 2: also a per-update dependency trace, PPCSR_DIAG_DUMP = file), "marker" (marker kernels for profile cuts), "test_block_rebalance" */
int ppcsr_set_option(ppcsr_t h, const char *key, int64_t value);
/* device-side copy of the whole state and return to it (used by the benchmark to replay a batch on the same
 * core graph, and by the engine itself as the rollback point of speculative rounds); no reference equivalent.
 * After the first copy both directions are incremental: only leaves / node records written since (dirty tags) move */
int ppcsr_snapshot(ppcsr_t h);
int ppcsr_restore(ppcsr_t h);
/* debugging / measurement helpers */
int ppcsr_check_invariants(ppcsr_t h, uint64_t *bad_leaves);
int ppcsr_bench_scan_all(ppcsr_t h, double *ms, uint64_t *total);
int ppcsr_bench_rebalance(ppcsr_t h, uint64_t window_slots, int iters, double *ms_per_call);
/* PCSR::double_list / half_list (PCSR.cpp:251-282, 284-320) alone: the array is doubled and halved back `iters` times; device time
 * per call of each (measurement helper: the array ends at its original size, evenly spread) */
int ppcsr_bench_resize(ppcsr_t h, int iters, double *double_ms, double *half_ms);
const char *ppcsr_strerror(int status);
const char *ppcsr_last_error(void); /* message of the last failing call on this thread */
int ppcsr_device_count(void);

/* ---- PPPCSR: vertex-range partitioning (PPPCSR.cpp:13-34, 58-66); one partition per GPU ------------------- */
/* PPPCSR::PPPCSR(init_n, src_n, lock_search, numDomain, partitionsPerDomain, use_numa): partition p lives on
 * devices[p % n_devices] (n_devices may be 1: all partitions on one GPU) */
int pppcsr_create(uint32_t init_n, uint32_t src_n, int lock_search, int num_domains, int parts_per_domain,
                  const int *devices, int n_devices, pppcsr_t *out);
/* the same layout, but only partitions [first_part, first_part + n_local_parts) are created, all on `device`: the form a
 * multi-process run uses (one rank per GPU holding the partitions of its domain; the other partitions live in other
 * processes and calls that route to them fail with EINVAL) */
int pppcsr_create_local(uint32_t init_n, int lock_search, int num_domains, int parts_per_domain, uint64_t first_part,
                        uint64_t n_local_parts, int device, pppcsr_t *out);
int pppcsr_destroy(pppcsr_t h);
int pppcsr_num_partitions(pppcsr_t h, uint64_t *out);
/* PPPCSR::get_partiton — PPPCSR.cpp:58-66 */
int pppcsr_get_partition(pppcsr_t h, uint64_t vertex, uint64_t *part);
int pppcsr_partition_start(pppcsr_t h, uint64_t part, uint64_t *first_vertex);
int pppcsr_partition(pppcsr_t h, uint64_t part, ppcsr_t *out); /* borrowed handle */
/* PPPCSR::add_edge / remove_edge / edge_exists / get_neighbourhood / getNode / get_n / add_node — PPPCSR.cpp:36-80 */
int pppcsr_add_edge(pppcsr_t h, uint32_t src, uint32_t dst, uint32_t value);
int pppcsr_remove_edge(pppcsr_t h, uint32_t src, uint32_t dst);
int pppcsr_edge_exists(pppcsr_t h, uint32_t src, uint32_t dst, int *exists);
int pppcsr_get_neighbourhood(pppcsr_t h, int src, int *out, uint64_t cap, uint64_t *count);
int pppcsr_get_node(pppcsr_t h, uint32_t v, ppcsr_node *out);
int pppcsr_get_n(pppcsr_t h, uint64_t *n);
int pppcsr_add_node(pppcsr_t h);
/* bucket a host stream by owner (stable: per-partition order == stream order, src made partition-local as in
 * PPPCSR.cpp:46-52) and apply each bucket on its partition's GPU */
int pppcsr_apply_batch(pppcsr_t h, const ppcsr_op *ops, uint64_t n);
/* the same for a batch already resident in HBM (all partitions on that one GPU): device-side stable bucketing, then every
 * partition applies its bucket on its own stream, driven by one host thread each — the reference's ThreadPoolPPPCSR runs
 * its domains' workers concurrently in the same way (thread_pool_pppcsr.cpp:121-156) */
int pppcsr_apply_batch_device(pppcsr_t h, const ppcsr_op *d_ops, uint64_t n);
/* already routed subsequences (partition-local src, stream order), one device pointer per partition of
 * [first_part, first_part + n_parts): what a rank holds after the all-to-all exchange; applied concurrently as above */
int pppcsr_apply_parts_device(pppcsr_t h, uint64_t first_part, uint64_t n_parts, const ppcsr_op *const *d_ops,
                              const uint64_t *counts);
/* owner-bucketing primitive for the multi-process (one rank per GPU) path: counts[p] = ops owned by p,
 * bucketed = ops stably grouped by owner with partition-local src.  Pure host routine. */
int pppcsr_bucket_ops(uint32_t init_n, uint64_t n_parts, const ppcsr_op *ops, uint64_t n, ppcsr_op *bucketed,
                      uint64_t *counts);

/* the same routing for a block of the stream already resident in HBM (multi-GPU exchange): stable counting sort by
 * owner on `stream` (a hipStream_t, may be NULL); d_counts[p] (device memory, n_parts <= 64 entries) receives the bucket sizes */
int pppcsr_bucket_ops_device(uint32_t init_n, uint64_t n_parts, const ppcsr_op *d_ops, uint64_t n, ppcsr_op *d_bucketed,
                             uint64_t *d_counts, void *stream);

/* ---- multi-GPU owner exchange: what replaces ThreadPoolPPPCSR::submit_* (thread_pool_pppcsr.cpp:96-118) across processes ----
 * One process per GPU; each holds the contiguous partition range [rank * P / ranks, (rank + 1) * P / ranks) of the global
 * layout (pppcsr_create_local) and a contiguous block of the global stream in HBM.  The exchange has three steps:
 *   pack      stable device-side bucketing of the block by owner partition (src made partition-local, PPPCSR.cpp:46-52).
 *             The bucketed block IS the send buffer: ranks hold ascending partition ranges, so the rows for peer r are
 *             the contiguous run of its partitions' buckets — nothing is padded, only `counts` rows travel.
 *   transport first the bucket sizes (ppr numbers to and from every peer), then the rows: segment (source r, partition q)
 *             lands directly where the stream of partition q wants it (partition-major, source-rank-minor = global
 *             stream order per partition) — no unpack pass.
 *   apply     every local partition applies its stream concurrently (pppcsr_apply_parts_device).
 * pppcsr_exchange_apply runs all three with RCCL as the carrier (grouped ncclSend / ncclRecv on a HIP stream, bound at run
 * time by dlopen of librccl.so; two small collective steps per batch).  The pppcsr_xchg_* calls expose the same pack /
 * layout / apply code with the transport left to the caller (another carrier, or the tests' gloo transport between two
 * CPU-emulator processes).  pppcsr_xchg_create fails unless P is a multiple of n_ranks and the partitions resident in `h`
 * are exactly this rank's range, on one device.
 * The 128-byte unique id comes from rank 0 (pppcsr_comm_unique_id) and is handed to the other ranks by the host program
 * (any bootstrap: a file, MPI, torch.distributed's store).  Every rank must call pppcsr_exchange_apply for every batch, also
 * with an empty block.  A rank whose bucketing fails tells its peers through the counts, and no rank exchanges rows. */
typedef struct pppcsr_xchg *pppcsr_xchg_t; /* staging of one rank's side of the exchange */
int pppcsr_xchg_create(pppcsr_t h, int n_ranks, int rank, pppcsr_xchg_t *out);
int pppcsr_xchg_destroy(pppcsr_xchg_t x);
/* send_counts[P] (host, may be NULL): rows per partition; partition p's rows start at *d_send + sum(send_counts[0..p)) */
int pppcsr_xchg_pack(pppcsr_xchg_t x, const ppcsr_op *d_ops, uint64_t n, uint64_t *send_counts, const ppcsr_op **d_send);
/* recv_counts[r * ppr + q] = rows source rank r holds for my q-th partition (ppr = P / n_ranks); d_dst[r * ppr + q]
 * (may be NULL) = device address that segment must be written to before pppcsr_xchg_apply */
int pppcsr_xchg_layout(pppcsr_xchg_t x, const uint64_t *recv_counts, ppcsr_op **d_dst);
int pppcsr_xchg_apply(pppcsr_xchg_t x);
int pppcsr_xchg_set_num_neighbors(pppcsr_xchg_t x); /* the routed records are (vertex, num_neighbors, *): see pppcsr_repartition_export */
int pppcsr_xchg_bulk_build(pppcsr_xchg_t x);        /* the routed records are the adds EMPTY partitions are bulk-built from: idem */
int pppcsr_comm_unique_id(void *id_out_128_bytes);
int pppcsr_comm_create(const void *id_128_bytes, int n_ranks, int rank, int device, pppcsr_comm_t *out);
int pppcsr_comm_destroy(pppcsr_comm_t c);
int pppcsr_exchange_apply(pppcsr_t h, pppcsr_comm_t c, const ppcsr_op *d_ops, uint64_t n);
int pppcsr_exchange_set_num_neighbors(pppcsr_t h, pppcsr_comm_t c, const ppcsr_op *d_recs, uint64_t n);
int pppcsr_exchange_bulk_build(pppcsr_t h, pppcsr_comm_t c, const ppcsr_op *d_adds, uint64_t n);

/* ---- repartitioning.  The reference only sketches it (PCSR.h:91-112 is commented out), so there
 * is no reference behaviour to match; the rule here is deterministic and the tests rebuild it with the oracle:
 * new_starts[P] = first global vertex of every partition (new_starts[0] = 0, non-decreasing).  A partition whose vertex
 * range is unchanged keeps its array as it is.  A partition whose range changes is recreated empty at its new size, and
 * the edges of all such partitions are returned as adds of the global stream — (global src, dest, value), ascending
 * (src, dest) — in device memory owned by the handle (valid until the next call).  They are routed like any batch (owner
 * bucketing by the new starts; the exchange across ranks, every rank calling with the same new_starts) into the BULK BUILD
 * of the recreated partitions (ppcsr_bulk_build's rules: a valid packed-memory array with these edges, not the layout
 * one-by-one inserts would leave — there is no reference layout here, and a source-sorted stream through the exact path
 * is the hot-vertex worst case): pppcsr_bulk_build_device in one process, pppcsr_exchange_bulk_build across ranks,
 * pppcsr_xchg_bulk_build with a carrier of your own.  num_neighbors is a counter of calls, not the degree (duplicate adds
 * and deletes of missing edges move it, PCSR.cpp:1380/1409), so it travels beside the edges: d_nn holds one record
 * (global vertex, num_neighbors, 1) per vertex of the changed partitions, routed the same way AFTER the build
 * (pppcsr_set_num_neighbors_device / pppcsr_exchange_set_num_neighbors / pppcsr_xchg_set_num_neighbors).
 * pppcsr_repartition does all of it when every partition is resident in this process.  Updates applied afterwards go
 * through the ordinary (sequentially exact) path.
 * pppcsr_balanced_starts proposes starts of about equal weight, weight(v) = num_neighbors(v) + 1. */
int pppcsr_repartition_export(pppcsr_t h, const uint64_t *new_starts, const ppcsr_op **d_ops, uint64_t *n, const ppcsr_op **d_nn,
                              uint64_t *n_nn);
int pppcsr_set_num_neighbors_device(pppcsr_t h, const ppcsr_op *d_recs, uint64_t n);
int pppcsr_bulk_build_device(pppcsr_t h, const ppcsr_op *d_adds, uint64_t n); /* global src; every receiving partition must be empty */
int pppcsr_repartition(pppcsr_t h, const uint64_t *new_starts);
int pppcsr_balanced_starts(pppcsr_t h, uint64_t *starts_out);

#ifdef __cplusplus
}
#endif
#endif
