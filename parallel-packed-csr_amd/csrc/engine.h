// Host side of the MI355X PMA engine: owns the HBM-resident state, drives the round scheduler and the
// exclusive executor, and implements the whole-array phases (double_list / half_list / big windows).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "pma_types.h"

namespace ppcsr {

struct View;  // pma_device.h
struct Edge;

struct EngineStats {
  uint64_t N, n;
  int logN, H;
  uint64_t rounds, committed, planned, exclusive_ops, round_syncs;
  uint64_t redistribute_calls, redistribute_slots;  // algorithmic (what the reference performs)
  uint64_t double_calls, half_calls, big_redistributes, rollbacks;
  uint64_t not_found, duplicates, noops, slide_slots;
  uint64_t ops_applied;
  double last_batch_ms;      // device-only time of the last apply_batch (ops resident in HBM)
  double last_batch_h2d_ms;  // time of the H2D copy of the op array (host-buffer entry point)
  // profile mode (option "profile"=1): HIP-event time of each round kernel on the engine's stream
  double prof_plan_ms, prof_check_ms, prof_apply_ms, prof_compact_ms;
  uint64_t prof_launches;  // launches of EACH of the three kernels
  uint64_t wasted_rounds;  // rounds of speculative epochs that were rolled back (not part of `rounds`)
  uint64_t narrow_lost;    // times add_node-after-doubling switched the structure to the sequential regime
  uint64_t narrow;         // 1: sorted, disjoint vertex ranges (64-ary search narrowing + parallel rounds); 0: sequential regime
};

class Engine {
 public:
  static int create(uint32_t init_n, uint32_t src_n, int lock_search, int device, Engine **out, std::string *errmsg = nullptr);
  ~Engine();

  int apply_batch_host(const Op *ops, uint64_t n);
  int apply_batch_device(const Op *d_ops, uint64_t n);
  int add_edge(uint32_t s, uint32_t d, uint32_t value);
  int remove_edge(uint32_t s, uint32_t d);
  int add_node();
  int edge_exists(uint32_t s, uint32_t d, int *out);
  int get_node(uint32_t v, Node *out);
  int get_neighbourhood(int src, int *out, uint64_t cap, uint64_t *count);
  int read_neighbourhood(int src);
  int scan_all(uint64_t *row_offsets, int *dests, uint64_t cap, uint64_t *total);
  int export_triples_device(uint32_t src_base, Op *d_out, uint64_t cap, uint64_t *total);  // edges as adds of the global stream
  int export_num_neighbors_device(uint32_t base, Op *d_out);       // n records (vertex + base, num_neighbors, 1)
  int set_num_neighbors_device(const Op *d_recs, uint64_t cnt);    // records (local vertex, num_neighbors, *)
  int scan_all_device(double *ms, uint64_t *total);  // device-only timing of the bulk scan (bench)
  // graph-algorithm consumers over the gapped array (reference: src/utility/bfs.h, src/utility/pagerank.h)
  int bulk_build(const Op *host_ops, uint64_t m, double *device_ms);  // non-parity fast path
  int bulk_build_device(const Op *d_adds, uint64_t m, double *device_ms);  // the same, adds already in HBM (pppcsr_repartition)
  int bfs(uint32_t start, uint32_t *levels, double *device_ms);
  int pagerank(const float *node_values, float *out, double *device_ms);
  int export_state(Edge *items, Node *nodes);
  int check_invariants(uint64_t *bad);  // leafcnt == recount(items)
  int stats(EngineStats *out);
  int set_option(const char *key, int64_t value);
  int rebalance_bench(uint64_t wlen, int iters, double *ms_per_call);
  int resize_bench(int iters, double *double_ms, double *half_ms);
  int snapshot();  // device-side copy of the whole state (items, nodes, leaf counts, geometry)
  int restore();   // back to the last snapshot (device-to-device)  // whole-window rebalance kernel timing

  uint64_t N() const;
  uint32_t n() const;
  int logN() const;
  int H() const;
  int device() const { return device_; }
  const std::string &last_error() const { return err_; }

 private:
  Engine();
  int init(uint32_t init_n, uint32_t src_n, int lock_search, int device);
  int run_rounds(const Op *d_ops, uint64_t n);
  // spec_index != kMax: the update sits at that index of d_ops and runs inside a speculative epoch (stamp-validated);
  // *violation / *resized report what happened
  int run_exclusive(Op op, uint32_t flags, const Op *d_ops = nullptr, uint32_t spec_index = 0xFFFFFFFFu, bool *violation = nullptr,
                    bool *resized = nullptr, bool later_committed = false);
  int resize(uint64_t newN);
  int big_redistribute(uint64_t wstart, uint64_t wlen, bool sync = true);
  int inplace_fault_check();
  int rank_scan(const uint32_t *d_cnt, uint64_t nleaves, bool table = false, uint64_t tb_index = 0, uint64_t tb_len = 0);
  int recheck_ranges();  // narrow == 0: look whether the vertex ranges are sane again and re-enable narrowing + parallel rounds
  int ensure_scratch(uint64_t nleaves);
  int ensure_tiles(uint64_t ntiles);
  int fail(int code, const std::string &msg);
  int pull_stats();

  int run_speculative(const Op *d_ops, uint64_t n);
  int rebalance_fused(const View &nv, const Edge *src_items, uint64_t src_lo, uint64_t src_len, int src_sh, uint32_t *src_cnt,
                      bool inplace, uint64_t tb_index, uint64_t tb_len, Edge *dst, uint64_t dst_bias, uint32_t *dst_cnt, uint64_t dst_nleaves);
  int bulk_build_from(const Op *in_ops, bool on_device, uint64_t m, double *device_ms);
  int scan_launch(unsigned long long *d_rows, int *d_dst, uint64_t cap, const float *d_values = nullptr, float *d_contrib = nullptr,
                  Op *d_triples = nullptr, uint32_t src_base = 0);

 public:
  struct Impl;

 private:
  Impl *p_;
  int device_ = 0;
  std::string err_;
};

const char *error_string(int code);
// stable owner bucketing of a device-resident block of the stream (multi-GPU exchange); runs on `stream`
// (starts: the first global vertex of each of the n_parts <= 64 partitions, host memory)
int bucket_ops_device(const uint32_t *starts, uint32_t n_parts, const Op *d_ops, uint64_t n, Op *d_out, unsigned long long *d_counts,
                      void *stream, std::string *errmsg);
}  // namespace ppcsr
int gpu_device_count_for_capi(int *n);
namespace ppcsr {

}  // namespace ppcsr
