// Host-side driver of the PMA engine (see engine.h).  Compiled by hipcc into libppcsr_hip.so.
#include "engine.h"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <memory>
#include <vector>
#include <chrono>

#include "gpu_rt.h"
#define GPU_DFREE(x) gpu::dfree_named((x), #x, __LINE__)
#include "pma_kernels.h"

namespace ppcsr {

#define GCHK(expr)                                                                       \
  do {                                                                                   \
    int _e = (expr);                                                                     \
    if (_e != 0) return fail(PPCSR_EHIP, std::string(#expr) + ": " + gpu::err_str(_e)); \
  } while (0)

// frees the device buffers registered with it on every exit path (error returns included) unless dismissed
struct DevGuard {
  std::vector<void **> slots;
  template <class T>
  void add(T **p) { slots.push_back(reinterpret_cast<void **>(p)); }
  void dismiss() { slots.clear(); }
  ~DevGuard() {
    for (void **q : slots)
      if (*q) {
        gpu::dfree(*q);
        *q = nullptr;
      }
  }
};

const char *error_string(int code) {
  switch (code) {
    case PPCSR_OK: return "ok";
    case PPCSR_EINVAL: return "invalid argument";
    case PPCSR_ENOMEM: return "out of memory";
    case PPCSR_EHIP: return "HIP runtime error";
    case PPCSR_EUNSUPPORTED: return "unsupported rare path (no null slot on either side of a slide: completely full array)";
    case PPCSR_EINTERNAL: return "internal error";
    case PPCSR_ERANGE: return "output buffer too small";
    default: return "unknown error";
  }
}

struct Engine::Impl {
  View v{};
  gpu::stream_t stream{};
  uint64_t n_cap = 0;        // capacity of nodes[]
  uint64_t leaves_cap = 0;   // capacity of leafcnt/wres
  Control *d_ctl = nullptr, *h_ctl = nullptr;
  Plan *d_plans = nullptr;
  uint64_t plans_cap = 0;  // records in d_plans (shared by the strict and the speculative rounds)
  StatShard *d_stats = nullptr, *h_stats = nullptr, *d_stats_snap = nullptr;
  ExclOut *d_xout = nullptr, *h_xout = nullptr;
  Op *d_ops = nullptr;
  uint64_t ops_cap = 0;
  Op *h_op1 = nullptr;  // pinned staging for single ops
  // scan scratch
  uint32_t *d_rank = nullptr, *d_tiles = nullptr;
  uint64_t rank_cap = 0, tiles_cap = 0;
  unsigned long long *d_total = nullptr, *h_total = nullptr;
  ChainTable *d_table = nullptr;
  unsigned long long *d_scan_state = nullptr;
  uint64_t scan_state_cap = 0, scan_nchunks = 0;
  Edge *d_scratch = nullptr;  // persistent destination of big in-place rebalances
  uint64_t scratch_cap = 0;
  int *d_nbr = nullptr;
  uint64_t nbr_cap = 0;
  uint32_t round = 0;
  uint32_t max_horizon = 4096, min_horizon = 64, rounds_per_sync = 32, init_horizon = 256;
  gpu::Timer timer;
  EngineStats st{};
  // snapshots: second copies of the state in HBM.  `snap` = user snapshot()/restore(); `esnap` = rollback
  // point of the current speculative epoch
  struct Snap {
    View v{};
    uint64_t cap_slots = 0, cap_nodes = 0;
    bool valid = false;
    uint32_t synced = 0;   // serial at which this copy was last synchronised with the live state (dirty tags above it differ)
    uint64_t gen = 0;      // array generation it belongs to (a resize / bulk build / add_node starts a new one: full copy)
  };
  Snap snap, esnap;
  // dirty tags: writers stamp View::ldirty / vdirty with `serial`; every snapshot synchronisation advances it
  uint32_t serial = 1;
  uint64_t array_gen = 1;
  bool stamps_clean = false;  // the validation stamps hold nothing from an earlier batch or a rolled-back epoch
  // speculative scheduler state
  OptCtl *d_octl = nullptr, *h_octl = nullptr;
  uint32_t *d_vdbg = nullptr;
  uint32_t *d_status = nullptr, *d_carry0 = nullptr, *d_carry1 = nullptr;
  uint64_t carry_cap = 0, hslot_cap = 0;
  unsigned long long *d_regfail = nullptr, *d_pfail = nullptr;
  uint32_t *d_vws = nullptr, *d_vrs = nullptr;  // per-vertex sentinel stamps (capacity n_cap + 1)
  uint32_t *d_wstamp = nullptr, *d_rstamp = nullptr;
  uint32_t mode = 1;             // 0 = strict prefix rounds, 1 = speculative rounds with validated rollback
  uint32_t epoch_ops = 1u << 20;  // rollback granularity (upper bound)
  // adaptive epoch length: a rollback throws away everything since the epoch's snapshot, so a stream that provokes
  // rollbacks (hot vertices whose windows balloon while they wait) is cut into short epochs — kEpochShort updates after a
  // rollback, doubling again with every epoch that ends cleanly — while a stream that never rolls back keeps one
  // snapshot per epoch_ops updates
  uint32_t cur_epoch = 0;
  // (a snapshot costs about one round; a rollback re-does half an epoch on average.  On a config #4 partition — 3 rollbacks
  // per 1.25 M inserts — 8192 / 8 spent 16 % of the batch copying snapshots: 16384 / 2 took 30.4 ms instead of 38.4; the
  // Zipf stream on the same partition, 50 rollbacks per 1.5 M updates, was indifferent: 208 ms either way)
  uint32_t epoch_short = 16384;  // epoch length after a rollback
  uint32_t epoch_clean = 0, epoch_grow_after = 2;  // clean epochs in a row / how many of them double the length again
  // Round 3: with incremental snapshots an epoch boundary costs ~30 us, so a stream that rolls back often is better off with
  // epochs a good deal shorter than the distance between its rollbacks (hot-vertex stream: one rollback per 28 K updates;
  // 2048-update epochs that double after 4 clean ones: 121 -> 105 ms per 1 M), while a stream that rolls back three times
  // per million (a config #4 partition) needs the long ones (4096-update epochs: 24 -> 35 ms).  epoch_adapt = 1: the epoch
  // after a rollback is 1/epoch_adapt (default 8) of the running mean distance between rollbacks, within [2048, epoch_short].
  uint32_t epoch_adapt = 8, grow_eff = 2;  // (epoch_adapt: 0 off, else the divisor)
  uint64_t since_rollback = 0;
  double rb_dist = -1.0;
  // per-region prefix rule: nothing later may commit in a region where an earlier update was deferred.  4096 slots blocked
  // 43 K of the 69 K non-commits per 1 M updates of config #2 for nothing (179 M/s; 2048: 184, 1024: 187, 512: 188, 256: 189);
  // the hot-vertex stream needs the rule (1024: 142 ms as with 4096, 83 rollbacks instead of 36; 256: 191 ms, 199 rollbacks)
  uint32_t region_slots = 1024;
  // ... so the rule is tightened where rollbacks happen: after one, regions are region_wide slots until region_calm epochs in
  // a row have ended cleanly (a config #4 partition at critical density: 31.1 ms / 9 rollbacks with 1024 throughout, 26.7 ms
  // / 3 rollbacks with 4096; its calm second half is config-#2-like again)
  uint32_t region_wide = 4096, region_calm = 8, region_eff = 0, region_clean = 0;
  // Round 3: a stream whose rollbacks are RARE (running mean distance >= region_rare_dist updates: a config #4 partition at
  // critical density, one per ~150 K updates while its levels sit at their bounds) is better off paying for none at all:
  // region_rare slots after a rollback, for region_rare_calm (8) x that distance of committed updates (24.9 -> 19.3 ms per batch of
  // the partition: 193 -> 157 rounds, 0 rollbacks; the wide rule ends about where the array doubles and the stream turns calm).
  // A stream that rolls back all the time (hot vertices, one per 28 K updates) keeps region_wide / region_calm: with 16 K-slot
  // regions it loses more commits per round than the rollbacks cost (105 -> 131 ms); a stream that never rolls back never
  // sees either rule.
  uint32_t region_rare = 16384, region_rare_calm = 8, region_rare_dist = 65536;
  uint64_t region_rare_span = 0;  // > 0: the rare rule is on until this many updates have committed since the rollback
  // ... and only for streams that commit a good share of a chip-full per round (running mean >= region_rare_cpr): wide regions
  // make rollbacks rarer for ANY stream, so the distance alone would also switch the rule on for the hot-vertex stream once it
  // has been on for a while (600-750 commits per round there, 6-8 K on the critical-density partition)
  uint32_t region_rare_cpr = 1536;
  double cpr_mean = -1.0;
  // Round width (upper bound when `adaptive` is on).  One update = one wave; `resident_waves` of them fit the chip at once
  // (rounds 1-3: o_plan at 78 VGPRs = 6 waves per SIMD, 24 per CU, 6144 on 256 CUs; since round 4 o_plan runs 8 per SIMD and
  // o_apply — one 6 KB LDS tile per wave — 6: the unit is 7 per SIMD, 7168 on 256 CUs, measured best of 6144 / 7168 / 8192:
  // config #2 258 / 263 / 261 M/s).  A round's kernels are bound by latency, so a round
  // of 2 x resident takes ~1.4x the time of one of 1 x resident; widths in between leave the second pass partly empty
  // (config #2, updates/s: 6144 -> 139 M, 8192 -> 130 M, 12288 -> 162 M, 18432 -> 169 M, 24576 -> 153 M: the re-planned
  // share grows with the width — 3 %, 7 %, 12 %).  init() sets opt_horizon = 3 x resident, start_horizon = resident; the
  // adaptive width only climbs above 1 x while more than 85 % of a round commits.
  uint32_t resident_waves = 0;  // 0: unknown (emulator) — no quantisation of the adapted width
  uint32_t opt_horizon = 6144;    // (the CPU emulator, which reports no CUs, keeps these)
  uint32_t start_horizon = 6144;
  uint32_t adaptive = 1;
  uint32_t rb_defer_table = 1u << 22;  // windows from slot 0 of at least this many slots build their position table inside the
                                       // scatter launch (0: never — the table is built in front of it)
  uint32_t scatter_blocks = 8192;
  uint32_t small_batch = 256;    // batches up to this size take the strict rounds even in speculative mode
  // windows up to this size are rebalanced by the exclusive executor's own wave (64 slots at a time: ~2 us per dependent
  // chunk, i.e. milliseconds at 64 K slots); larger ones by the multi-workgroup kernels (three launches whatever the size)
  uint32_t excl_in_wave = 4096;
  // speculative rounds: windows up to big_window slots stay inside the round; those above big_min are rebalanced by a
  // workgroup each (o_big, big_grid workgroups, one scratch stretch of big_window slots per workgroup)
  uint32_t big_window = 32768, big_min = 256, big_grid = 64;
  dev::BigJob *d_jobs = nullptr;
  Edge *d_bigscratch = nullptr;
  uint64_t bigscratch_cap = 0;
  uint32_t rb_tile = 0;          // leaves per rebalance tile (power of two <= 256); 0 = pick per window
  uint32_t rb_min_tiles = 4096;  // auto tile: shrink the tile until the window has at least this many
  // o_big (the launch that rebalances the windows a round queues for workgroups) is left out of the rounds while a stream queues
  // none: a round that does queue one then stops the chunk (OptCtl::need_big), the host runs the launch and keeps it in for a while
  bool big_on = false;
  double big_rate = 0;
  // o_check: a lane per update while updates have short footprints (one leaf, two or three read ranges); a wave per update for
  // streams where more than ~1 in 64 has a long one (hot ranges, critical density: big windows, many-level climbs, runs of moved
  // sentinels), which the lane kernel leaves to its wave one by one
  bool check_lanes = true;
  uint32_t check_wave_chunks = 0;
  int check_force = -1;  // option "check_lanes": -1 by the share of long footprints (default), 0 always a wave per update, 1 always a lane
  uint32_t rb_prefetch = 1;  // 1: four chunks in flight per wave, 0: one
  bool time_resize = false;  // resize_bench: time the passes of resize() with events
  double last_resize_ms = 0;
  uint32_t rb_bench_upper = 0;  // rebalance_bench: 1 = the window [N - wlen, N) instead of [0, wlen)
  uint64_t rb_inplace_min = 1ull << 19;  // partial windows of at least this many slots are rebalanced in place (0 = never)
  uint32_t rb_inplace_cpw = 0;   // 64-slot chunks per wave of an in-place tile (8 or 16; 0 = by window size)
  uint32_t *d_ip = nullptr;      // in-place rebalance: header (sticky error, ticket counters), tile order, the tiles' flags
  uint32_t ip_epoch = 0;
  uint32_t ip_lists = 1;         // ticket lists of the in-place rebalance: the XCD ids seen at creation (k_xcc_probe), else 1
  bool ip_used = false;          // an in-place rebalance ran since the error flag was last looked at
  uint32_t scatter_variant = 2;  // 0: LDS-staged k_scatter_fill, 1: register-run k_scatter_runs, 2: runs + in-tile leaf scan (3 launches)
  bool partial = false;
  bool profile = false;  // bracket every round kernel with HIP events on the engine's stream
  bool in_batch = false;  // inside apply_batch_device (which ends with inplace_fault_check)
  // slots; a planned window at least this big lets nothing later overtake it (0: big_window / 2).  Such an update is
  // likely to turn exclusive once the earlier updates have landed, and what has overtaken it by then is rolled back — but
  // everything behind the barrier waits a round: at big_window / 4 a config #4 partition (critical density: a window of
  // 8192 slots every few thousand updates) had 67 % of its non-commits "behind a barrier" and ran 373 rounds per 1.25 M
  // inserts; at / 2: 276 rounds and the same 2-3 rollbacks; without any: 234 rounds, but slower ones
  uint32_t soft_barrier = 0;
  uint32_t defer_barrier = 0;  // slots; a deferred update with a window at least this big lets nothing later overtake it (0: off)
  uint32_t dbg_repeat = 0;  // measurement aid (diagnostics build of the round kernels only): o_plan executes parts of a plan twice — see OptArgs::dbg
  uint32_t diag = 0;     // count, per epoch, why planned updates did not commit (printed to stderr at the end of the epoch)
  uint32_t *d_dg = nullptr;  // diag >= 2: per-update trace of the batch (OptArgs::dg)
  uint64_t dg_cap = 0;
  std::vector<gpu::Event> events;  // init failed half-way: destructor frees only what exists
};

Engine::Engine() : p_(new Impl()) {}

// d_plans is shared by both schedulers: it must hold one record per wave of the widest grid either may launch
// (+8: the per-wave arrays are padded to the launch grid, whose waves load their record before the early-exit tests)
static int ensure_plans(Engine::Impl &p) {
  const uint64_t need = (uint64_t)std::max(p.opt_horizon, p.max_horizon) + 64;  // (padded to the launch grids: 4 / 64 slots per workgroup)
  if (p.plans_cap >= need) return 0;
  int e = gpu::sync(p.stream);
  if (e) return e;
  Plan *np = nullptr;
  if ((e = gpu::dmalloc((void **)&np, need * sizeof(Plan)))) return e;
  if (p.d_plans) GPU_DFREE(p.d_plans);
  p.d_plans = np;
  p.plans_cap = need;
  return 0;
}

// every round-tagged reservation array restarts together with the round counter (stale keys must never meet a reused tag)
static int reset_tags(Engine::Impl &p) {
  const uint64_t leaves = p.v.g.N >> p.v.g.sh;
  int e;
  if (p.v.wres && (e = gpu::dset(p.v.wres, 0xFF, leaves * sizeof(unsigned long long), p.stream))) return e;
  if (p.v.rres && (e = gpu::dset(p.v.rres, 0xFF, leaves * sizeof(unsigned long long), p.stream))) return e;
  if (p.v.dres && (e = gpu::dset(p.v.dres, 0xFF, leaves * sizeof(unsigned long long), p.stream))) return e;
  if (p.d_regfail && (e = gpu::dset(p.d_regfail, 0xFF, (leaves + 1) * sizeof(unsigned long long), p.stream))) return e;
  if (p.d_pfail && (e = gpu::dset(p.d_pfail, 0xFF, (leaves + 1) * sizeof(unsigned long long), p.stream))) return e;
  if (p.v.vw && (e = gpu::dset(p.v.vw, 0xFF, (p.n_cap + 1) * sizeof(unsigned long long), p.stream))) return e;
  if (p.v.vr && (e = gpu::dset(p.v.vr, 0xFF, (p.n_cap + 1) * sizeof(unsigned long long), p.stream))) return e;
  p.round = 0;
  return 0;
}

// per-leaf auxiliary arrays that follow the geometry of `v` (reservations, stamps, region fail-mins)
static int alloc_aux(Engine::Impl &p, View &v) {
  const uint64_t leaves = v.g.N >> v.g.sh;
  int e;
  if ((e = gpu::dmalloc((void **)&v.wres, leaves * sizeof(unsigned long long)))) return e;
  if ((e = gpu::dmalloc((void **)&v.rres, leaves * sizeof(unsigned long long)))) return e;
  if ((e = gpu::dmalloc((void **)&v.dres, leaves * sizeof(unsigned long long)))) return e;
  if ((e = gpu::dset(v.dres, 0xFF, leaves * sizeof(unsigned long long), p.stream))) return e;
  if ((e = gpu::dmalloc((void **)&p.d_wstamp, leaves * sizeof(uint32_t)))) return e;
  if ((e = gpu::dmalloc((void **)&p.d_rstamp, leaves * sizeof(uint32_t)))) return e;
  if ((e = gpu::dmalloc((void **)&p.d_regfail, (leaves + 1) * sizeof(unsigned long long)))) return e;
  if ((e = gpu::dmalloc((void **)&p.d_pfail, (leaves + 1) * sizeof(unsigned long long)))) return e;
  if ((e = gpu::dset(p.d_pfail, 0xFF, (leaves + 1) * sizeof(unsigned long long), p.stream))) return e;
  if ((e = gpu::dset(v.wres, 0xFF, leaves * sizeof(unsigned long long), p.stream))) return e;
  if ((e = gpu::dset(v.rres, 0xFF, leaves * sizeof(unsigned long long), p.stream))) return e;
  if ((e = gpu::dset(p.d_regfail, 0xFF, (leaves + 1) * sizeof(unsigned long long), p.stream))) return e;
  if ((e = gpu::dset(p.d_wstamp, 0, leaves * sizeof(uint32_t), p.stream))) return e;
  if ((e = gpu::dset(p.d_rstamp, 0, leaves * sizeof(uint32_t), p.stream))) return e;
  if ((e = gpu::dmalloc((void **)&v.ldirty, leaves * sizeof(uint32_t)))) return e;
  if ((e = gpu::dset(v.ldirty, 0, leaves * sizeof(uint32_t), p.stream))) return e;
  p.leaves_cap = leaves;
  p.round = 0;
  if (v.vw && (e = gpu::dset(v.vw, 0xFF, (p.n_cap + 1) * sizeof(unsigned long long), p.stream))) return e;
  if (v.vr && (e = gpu::dset(v.vr, 0xFF, (p.n_cap + 1) * sizeof(unsigned long long), p.stream))) return e;
  return 0;
}
// per-vertex sentinel reservation / stamp arrays (follow the capacity of nodes[])
static int alloc_vertex_aux(Engine::Impl &p, View &v) {
  const uint64_t cap = p.n_cap + 1;
  int e;
  if (v.vw) GPU_DFREE(v.vw);
  if (v.vr) GPU_DFREE(v.vr);
  if (p.d_vws) GPU_DFREE(p.d_vws);
  if (p.d_vrs) GPU_DFREE(p.d_vrs);
  if (v.vdirty) GPU_DFREE(v.vdirty);
  v.vw = v.vr = nullptr;
  v.vdirty = nullptr;
  p.d_vws = p.d_vrs = nullptr;
  if ((e = gpu::dmalloc((void **)&v.vdirty, cap * sizeof(uint32_t)))) return e;
  if ((e = gpu::dset(v.vdirty, 0, cap * sizeof(uint32_t), p.stream))) return e;
  if ((e = gpu::dmalloc((void **)&v.vw, cap * sizeof(unsigned long long)))) return e;
  if ((e = gpu::dmalloc((void **)&v.vr, cap * sizeof(unsigned long long)))) return e;
  if ((e = gpu::dmalloc((void **)&p.d_vws, cap * sizeof(uint32_t)))) return e;
  if ((e = gpu::dmalloc((void **)&p.d_vrs, cap * sizeof(uint32_t)))) return e;
  if ((e = gpu::dset(v.vw, 0xFF, cap * sizeof(unsigned long long), p.stream))) return e;
  if ((e = gpu::dset(v.vr, 0xFF, cap * sizeof(unsigned long long), p.stream))) return e;
  if ((e = gpu::dset(p.d_vws, 0, cap * sizeof(uint32_t), p.stream))) return e;
  if ((e = gpu::dset(p.d_vrs, 0, cap * sizeof(uint32_t), p.stream))) return e;
  return 0;
}
static void free_aux(Engine::Impl &p, View &v) {
  if (v.ldirty) GPU_DFREE(v.ldirty);
  v.ldirty = nullptr;
  GPU_DFREE(v.wres);
  GPU_DFREE(v.rres);
  GPU_DFREE(v.dres);
  v.dres = nullptr;
  GPU_DFREE(p.d_wstamp);
  GPU_DFREE(p.d_rstamp);
  GPU_DFREE(p.d_regfail);
  GPU_DFREE(p.d_pfail);
  p.d_pfail = nullptr;
  v.wres = v.rres = nullptr;
  p.d_wstamp = p.d_rstamp = nullptr;
  p.d_regfail = nullptr;
}

int Engine::fail(int code, const std::string &msg) {
  err_ = msg;
  return code;
}

uint64_t Engine::N() const { return p_->v.g.N; }
uint32_t Engine::n() const { return p_->v.g.n; }
int Engine::logN() const { return p_->v.g.logN; }
int Engine::H() const { return p_->v.g.H; }

static inline uint32_t grid_for(uint64_t work_items, uint32_t per_block, uint32_t cap = 2048 * 4) {
  uint64_t b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (uint32_t)b;
}

int Engine::create(uint32_t init_n, uint32_t src_n, int lock_search, int device, Engine **out, std::string *errmsg) {
  Engine *e = new Engine();
  int rc = e->init(init_n, src_n, lock_search, device);
  if (rc != PPCSR_OK) {
    if (errmsg) *errmsg = e->err_;
    e->p_->partial = true;
    delete e;
    *out = nullptr;
    return rc;
  }
  *out = e;
  return PPCSR_OK;
}

int Engine::init(uint32_t init_n, uint32_t src_n, int lock_search, int device) {
  Impl &p = *p_;
  device_ = device;
  int ndev = 0;
  GCHK(gpu::device_count(&ndev));
  if (ndev <= 0) return fail(PPCSR_EHIP, "no HIP device visible: the MI355X engine has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(PPCSR_EINVAL, "bad device ordinal");
  GCHK(gpu::set_device(device));
  (void)gpu::last_error();  // drop any stale error left on this thread by earlier, unrelated runtime calls
  GCHK(gpu::stream_create(&p.stream));
  GCHK(p.timer.init());
  {
    int cus = 0;
    GCHK(gpu::device_cus(device, &cus));
    if (cus > 0) {
      p.resident_waves = (uint32_t)cus * 28u;
      p.start_horizon = p.resident_waves;
      p.opt_horizon = 3u * p.resident_waves;
    }
  }
  const uint64_t N = initial_N(init_n, src_n);
  Geometry g;
  compute_geometry(N, src_n, lock_search, &g);
  p.v.g = g;
  p.v.big_window = kBigWindow;
  p.v.serial = p.serial;
  p.n_cap = std::max<uint64_t>(src_n, 16);
  p.leaves_cap = N >> g.sh;
  GCHK(gpu::dmalloc((void **)&p.v.items, N * sizeof(Edge)));
  GCHK(gpu::dmalloc((void **)&p.v.nodes, p.n_cap * sizeof(Node)));
  GCHK(gpu::dmalloc((void **)&p.v.leafcnt, p.leaves_cap * sizeof(uint32_t)));
  GCHK(alloc_aux(p, p.v));
  GCHK(alloc_vertex_aux(p, p.v));
  GCHK(gpu::dmalloc((void **)&p.d_octl, sizeof(OptCtl)));
  GCHK(gpu::hmalloc((void **)&p.h_octl, sizeof(OptCtl)));
  GCHK(gpu::dmalloc((void **)&p.d_ctl, sizeof(Control)));
  GCHK(gpu::hmalloc((void **)&p.h_ctl, sizeof(Control)));
  GCHK(gpu::dmalloc((void **)&p.d_stats, kStatShards * sizeof(StatShard)));
  GCHK(gpu::hmalloc((void **)&p.h_stats, kStatShards * sizeof(StatShard)));
  GCHK(gpu::dset(p.d_stats, 0, kStatShards * sizeof(StatShard), p.stream));
  GCHK(gpu::dmalloc((void **)&p.d_stats_snap, kStatShards * sizeof(StatShard)));
  GCHK(gpu::dmalloc((void **)&p.d_xout, sizeof(ExclOut)));
  GCHK(gpu::hmalloc((void **)&p.h_xout, sizeof(ExclOut)));
  GCHK(gpu::hmalloc((void **)&p.h_op1, sizeof(Op)));
  GCHK(gpu::dmalloc((void **)&p.d_total, 32 * sizeof(unsigned long long)));  // ([0]: the total; the rest: scratch of debugging builds)
  GCHK(gpu::hmalloc((void **)&p.h_total, sizeof(unsigned long long)));
  GCHK(gpu::dmalloc((void **)&p.d_table, sizeof(ChainTable)));
  GCHK(gpu::dmalloc((void **)&p.d_ip, (kIpHdrWords + 2 * (uint64_t)kIpMaxTiles) * sizeof(uint32_t)));
  GCHK(gpu::dset(p.d_ip, 0, (kIpHdrWords + 2 * (uint64_t)kIpMaxTiles) * sizeof(uint32_t), p.stream));
  {  // which XCD ids do workgroups report here?  (the in-place rebalance stripes its ticket counter over them)
    uint32_t *probe = p.d_ip + kIpHdrWords;  // (scratch: the order list is written before every use)
    GPU_LAUNCH(p.stream, k_xcc_probe, 2048, 64, probe);
    uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    GCHK(gpu::d2h(cnt, probe, sizeof(cnt), p.stream));
    GCHK(gpu::sync(p.stream));
    GCHK(gpu::dset(probe, 0, sizeof(cnt), p.stream));
    uint32_t L = 0, lo = ~0u, hi = 0;
    while (L < 8 && cnt[L]) L++;
    bool ok = (L == 1 || L == 2 || L == 4 || L == 8);
    for (uint32_t i = 0; i < 8; i++) {
      if (i >= L && cnt[i]) ok = false;  // (ids must be exactly 0 .. L-1)
      if (i < L) {
        lo = std::min(lo, cnt[i]);
        hi = std::max(hi, cnt[i]);
      }
    }
    p.ip_lists = (ok && lo * 2 >= hi) ? L : 1u;
  }
  GCHK(ensure_plans(p));
  memset(p.h_ctl, 0, sizeof(Control));

  // constructor layout (PCSR.cpp:796-837): sentinel positions come from an fp64 accumulator, O(n) on the host
  std::vector<Node> nodes(src_n);
  {
    double index_d = 0.0;
    const double step = ((double)N) / src_n;
    for (uint32_t i = 0; i < src_n; i++) {
      nodes[i].beginning = (i == 0) ? 0u : nodes[i - 1].end;
      index_d += step;
      nodes[i].end = (uint32_t)(int)index_d;
      nodes[i].num_neighbors = 0;
    }
    if (src_n != 0) nodes[src_n - 1].end = (uint32_t)(N - 1);
  }
  if (src_n) {
    GCHK(gpu::h2d(p.v.nodes, nodes.data(), (uint64_t)src_n * sizeof(Node), p.stream));
    GCHK(gpu::sync(p.stream));
  }
  GPU_LAUNCH(p.stream, k_fill_null, grid_for(N * 3, 256 * 8), 256, p.v.items, (uint64_t)0, N);
  if (src_n) GPU_LAUNCH(p.stream, k_place_sentinels, grid_for(src_n, 256), 256, p.v);
  GPU_LAUNCH(p.stream, k_recount, grid_for((N + 63) / 64, 4), 256, p.v, (uint64_t)0, N);
  GCHK(gpu::sync(p.stream));
  GCHK(gpu::last_error());
  return PPCSR_OK;
}

Engine::~Engine() {
  Impl &p = *p_;
  gpu::set_device(device_);
  gpu::sync(p.stream);
  GPU_DFREE(p.v.items);
  GPU_DFREE(p.v.nodes);
  GPU_DFREE(p.v.leafcnt);
  free_aux(p, p.v);
  if (p.v.vw) GPU_DFREE(p.v.vw);
  if (p.v.vr) GPU_DFREE(p.v.vr);
  if (p.d_vws) GPU_DFREE(p.d_vws);
  if (p.d_vrs) GPU_DFREE(p.d_vrs);
  if (p.v.vdirty) GPU_DFREE(p.v.vdirty);
  GPU_DFREE(p.d_octl);
  gpu::hfree(p.h_octl);
  if (p.d_status) GPU_DFREE(p.d_status);
  if (p.d_vdbg) GPU_DFREE(p.d_vdbg);
  if (p.d_carry0) GPU_DFREE(p.d_carry0);
  if (p.d_carry1) GPU_DFREE(p.d_carry1);
  GPU_DFREE(p.d_ctl);
  gpu::hfree(p.h_ctl);
  GPU_DFREE(p.d_stats);
  gpu::hfree(p.h_stats);
  GPU_DFREE(p.d_stats_snap);
  GPU_DFREE(p.d_xout);
  gpu::hfree(p.h_xout);
  gpu::hfree(p.h_op1);
  GPU_DFREE(p.d_total);
  gpu::hfree(p.h_total);
  GPU_DFREE(p.d_table);
  if (p.d_ip) GPU_DFREE(p.d_ip);
  GPU_DFREE(p.d_plans);
  if (p.d_ops) GPU_DFREE(p.d_ops);
  if (p.d_rank) GPU_DFREE(p.d_rank);
  if (p.d_tiles) GPU_DFREE(p.d_tiles);
  if (p.d_nbr) GPU_DFREE(p.d_nbr);
  if (p.d_scratch) GPU_DFREE(p.d_scratch);
  if (p.d_scan_state) GPU_DFREE(p.d_scan_state);
  if (p.d_jobs) GPU_DFREE(p.d_jobs);
  if (p.d_bigscratch) GPU_DFREE(p.d_bigscratch);
  if (p.d_dg) GPU_DFREE(p.d_dg);
  for (Impl::Snap *sp : {&p.snap, &p.esnap}) {
    if (sp->v.items) GPU_DFREE(sp->v.items);
    if (sp->v.nodes) GPU_DFREE(sp->v.nodes);
    if (sp->v.leafcnt) GPU_DFREE(sp->v.leafcnt);
  }
  for (auto &e : p.events) e.destroy();
  p.timer.destroy();
  gpu::stream_destroy(p.stream);
  delete p_;
}

int Engine::set_option(const char *key, int64_t value) {
  Impl &p = *p_;
  std::string k(key ? key : "");
  if (k == "max_horizon") {
    if (value < 1 || value > (1 << 20)) return fail(PPCSR_EINVAL, "max_horizon out of range");
    gpu::set_device(device_);
    p.max_horizon = (uint32_t)value;
    GCHK(ensure_plans(p));
    if (p.min_horizon > p.max_horizon) p.min_horizon = p.max_horizon;
    if (p.init_horizon > p.max_horizon) p.init_horizon = p.max_horizon;
    return PPCSR_OK;
  }
  if (k == "min_horizon") {
    if (value < 1) return fail(PPCSR_EINVAL, "min_horizon out of range");
    p.min_horizon = (uint32_t)std::min<int64_t>(value, p.max_horizon);
    return PPCSR_OK;
  }
  if (k == "init_horizon") {
    if (value < 1) return fail(PPCSR_EINVAL, "init_horizon out of range");
    p.init_horizon = (uint32_t)std::min<int64_t>(value, p.max_horizon);
    return PPCSR_OK;
  }
  if (k == "mode") {
    if (value != 0 && value != 1) return fail(PPCSR_EINVAL, "mode must be 0 (strict) or 1 (speculative)");
    p.mode = (uint32_t)value;
    return PPCSR_OK;
  }
  if (k == "epoch_ops") {
    if (value < 1 || value > (1 << 24)) return fail(PPCSR_EINVAL, "epoch_ops out of range");
    p.epoch_ops = (uint32_t)value;
    return PPCSR_OK;
  }
  if (k == "region_slots") {
    if (value < 1 || (value & (value - 1))) return fail(PPCSR_EINVAL, "region_slots must be a power of two");
    p.region_slots = (uint32_t)value;
    p.region_eff = 0;
    return PPCSR_OK;
  }
  if (k == "region_wide") {
    if (value < 1 || (value & (value - 1))) return fail(PPCSR_EINVAL, "region_wide must be a power of two");
    p.region_wide = (uint32_t)value;
    return PPCSR_OK;
  }
  if (k == "region_rare") {  // 0: off
    if (value < 0 || (value & (value - 1))) return fail(PPCSR_EINVAL, "region_rare must be 0 or a power of two");
    p.region_rare = (uint32_t)value;
    return PPCSR_OK;
  }
  if (k == "region_rare_calm") {
    p.region_rare_calm = (uint32_t)std::max<int64_t>(1, value);
    return PPCSR_OK;
  }
  if (k == "region_rare_cpr") {
    p.region_rare_cpr = (uint32_t)std::max<int64_t>(0, value);
    return PPCSR_OK;
  }
  if (k == "region_rare_dist") {
    p.region_rare_dist = (uint32_t)std::max<int64_t>(1, value);
    return PPCSR_OK;
  }
  if (k == "region_calm") {
    p.region_calm = (uint32_t)std::max<int64_t>(1, value);
    return PPCSR_OK;
  }
  if (k == "opt_horizon") {
    if (value < 1 || value > (1 << 20)) return fail(PPCSR_EINVAL, "opt_horizon out of range");
    p.opt_horizon = (uint32_t)value;
    gpu::set_device(device_);
    GCHK(ensure_plans(p));
    p.start_horizon = std::min<uint32_t>(p.start_horizon, p.opt_horizon);
    if (!p.adaptive) p.start_horizon = p.opt_horizon;
    return PPCSR_OK;
  }
  if (k == "scatter_blocks") {
    p.scatter_blocks = (uint32_t)std::max<int64_t>(64, value);
    return PPCSR_OK;
  }
  if (k == "search_narrow") {  // 0: the literal binary walk only (what a structure add_node has corrupted falls back to)
    p.v.g.narrow = value ? 1u : 0u;
    return PPCSR_OK;
  }
  if (k == "small_batch") {
    p.small_batch = value < 0 ? 0u : (uint32_t)value;
    return PPCSR_OK;
  }
  if (k == "excl_in_wave") {
    p.excl_in_wave = (uint32_t)std::max<int64_t>(64, value);
    return PPCSR_OK;
  }
  if (k == "big_window") {  // <= kBigWindow: o_big is not launched at all (every window of a round is rebalanced by its own wave)
    if (value < 64 || (value & (value - 1))) return fail(PPCSR_EINVAL, "big_window must be a power of two >= 64");
    p.big_window = (uint32_t)std::min<int64_t>(value, 1 << 20);
    return PPCSR_OK;
  }
  if (k == "big_min") {
    p.big_min = (uint32_t)std::max<int64_t>(64, std::min<int64_t>(value, kBigWindow));
    return PPCSR_OK;
  }
  if (k == "big_grid") {
    p.big_grid = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(value, 256));
    return PPCSR_OK;
  }
  if (k == "rb_tile") {
    uint32_t t = 0;
    if (value > 0) for (t = 8; t < (uint32_t)value && t < kRbTile; t <<= 1) {}
    p.rb_tile = t;
    return PPCSR_OK;
  }
  if (k == "rb_min_tiles") {
    p.rb_min_tiles = value < 1 ? 1u : (uint32_t)value;
    return PPCSR_OK;
  }
  if (k == "rb_inplace_lists") {  // (test hook: 1, 2, 4 or 8 ticket lists whatever the probe saw)
    if (value != 1 && value != 2 && value != 4 && value != 8) return fail(PPCSR_EINVAL, "rb_inplace_lists must be 1, 2, 4 or 8");
    p.ip_lists = (uint32_t)value;
    return PPCSR_OK;
  }
  if (k == "rb_inplace_cpw") {
    p.rb_inplace_cpw = value >= 16 ? 16u : (value >= 8 ? 8u : 0u);
    return PPCSR_OK;
  }
  if (k == "rb_inplace_min") {
    p.rb_inplace_min = value < 0 ? 0ull : (uint64_t)value;
    return PPCSR_OK;
  }
  if (k == "check_lanes") {
    p.check_force = value < 0 ? -1 : (value ? 1 : 0);
    return PPCSR_OK;
  }
  if (k == "rb_bench_upper") {
    p.rb_bench_upper = value ? 1u : 0u;
    return PPCSR_OK;
  }
  if (k == "rb_prefetch") {
    p.rb_prefetch = value ? 1u : 0u;
    return PPCSR_OK;
  }
  if (k == "scatter_variant") {
    p.scatter_variant = value > 2 ? 2u : (uint32_t)value;
    return PPCSR_OK;
  }
  if (k == "adaptive") {
    p.adaptive = value != 0;
    return PPCSR_OK;
  }
  if (k == "resident_waves") {
    if (value < 0 || value > (1 << 20)) return fail(PPCSR_EINVAL, "resident_waves out of range");
    p.resident_waves = (uint32_t)value;
    return PPCSR_OK;
  }
  if (k == "start_horizon") {
    if (value < 1 || value > (1 << 20)) return fail(PPCSR_EINVAL, "start_horizon out of range");
    p.start_horizon = (uint32_t)value;
    return PPCSR_OK;
  }
  if (k == "test_block_rebalance") {  // test hook: value = wstart << 32 | wlen; rebalances that window with the workgroup routine
    const uint64_t ws = (uint64_t)value >> 32, wl = (uint64_t)value & 0xFFFFFFFFull;
    if (wl < 64 || (wl & (wl - 1)) || ws % wl || ws + wl > p.v.g.N || (wl >> p.v.g.sh) > dev::kBigLeaves) return fail(PPCSR_EINVAL, "bad window");
    gpu::set_device(device_);
    Edge *sc = nullptr;
    GCHK(gpu::dmalloc((void **)&sc, wl * sizeof(Edge)));
    GPU_LAUNCH(p.stream, k_block_rebalance, 1, dev::kBigThreads, p.v, ws, wl, sc);
    GCHK(gpu::sync(p.stream));
    GCHK(gpu::last_error());
    gpu::dfree(sc);
    return PPCSR_OK;
  }
  if (k == "marker") {
    gpu::set_device(device_);
    uint32_t *np_ = nullptr;
    switch (value) {
      case 0: GPU_LAUNCH(p.stream, k_mark_0, 1, 64, np_); break;
      case 1: GPU_LAUNCH(p.stream, k_mark_1, 1, 64, np_); break;
      case 2: GPU_LAUNCH(p.stream, k_mark_2, 1, 64, np_); break;
      case 3: GPU_LAUNCH(p.stream, k_mark_3, 1, 64, np_); break;
      case 4: GPU_LAUNCH(p.stream, k_mark_4, 1, 64, np_); break;
      case 5: GPU_LAUNCH(p.stream, k_mark_5, 1, 64, np_); break;
      case 6: GPU_LAUNCH(p.stream, k_mark_6, 1, 64, np_); break;
      case 7: GPU_LAUNCH(p.stream, k_mark_7, 1, 64, np_); break;
      default: return fail(PPCSR_EINVAL, "marker id must be 0..7");
    }
    return PPCSR_OK;
  }
  if (k == "epoch_short") {
    p.epoch_short = (uint32_t)std::max<int64_t>(64, value);
    return PPCSR_OK;
  }
  if (k == "epoch_adapt") {
    p.epoch_adapt = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(value, 1024));
    return PPCSR_OK;
  }
  if (k == "epoch_grow_after") {
    p.epoch_grow_after = (uint32_t)std::max<int64_t>(1, value);
    return PPCSR_OK;
  }
  if (k == "soft_barrier") {
    p.soft_barrier = (uint32_t)std::max<int64_t>(0, value);
    return PPCSR_OK;
  }
  if (k == "defer_barrier") {
    p.defer_barrier = (uint32_t)std::max<int64_t>(0, value);
    return PPCSR_OK;
  }
  if (k == "rb_defer_table") {
    p.rb_defer_table = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(value, 1ll << 31));
    return PPCSR_OK;
  }
  if (k == "dbg_repeat") {
    p.dbg_repeat = (uint32_t)value;
    return PPCSR_OK;
  }
  if (k == "diag") {  // 1: per-epoch counts of why updates did not commit; 2: + a per-update trace, dumped per epoch to $PPCSR_DIAG_DUMP
    p.diag = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(value, 2));
    return PPCSR_OK;
  }
  if (k == "profile") {
    p.profile = value != 0;
    p.st.prof_plan_ms = p.st.prof_check_ms = p.st.prof_apply_ms = p.st.prof_compact_ms = 0;
    p.st.prof_launches = 0;
    return PPCSR_OK;
  }
  if (k == "rounds_per_sync") {
    if (value < 1 || value > 4096) return fail(PPCSR_EINVAL, "rounds_per_sync out of range");
    p.rounds_per_sync = (uint32_t)value;
    return PPCSR_OK;
  }
  return fail(PPCSR_EINVAL, "unknown option " + k);
}

// ---- batch application ---------------------------------------------------------------------------------------
int Engine::apply_batch_host(const Op *ops, uint64_t n) {
  Impl &p = *p_;
  if (n == 0) return PPCSR_OK;
  if (!ops) return fail(PPCSR_EINVAL, "null ops");
  GCHK(gpu::set_device(device_));
  if (n > p.ops_cap) {
    if (p.d_ops) GPU_DFREE(p.d_ops);
    p.d_ops = nullptr;
    p.ops_cap = 0;
    GCHK(gpu::dmalloc((void **)&p.d_ops, n * sizeof(Op)));
    p.ops_cap = n;
  }
  auto t0 = std::chrono::steady_clock::now();
  GCHK(gpu::h2d(p.d_ops, ops, n * sizeof(Op), p.stream));
  GCHK(gpu::sync(p.stream));
  p.st.last_batch_h2d_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return apply_batch_device(p.d_ops, n);
}

int Engine::apply_batch_device(const Op *d_ops, uint64_t n) {
  Impl &p = *p_;
  if (n == 0) return PPCSR_OK;
  GCHK(gpu::set_device(device_));
  p.timer.start(p.stream);
  struct InBatch {
    bool &f;
    explicit InBatch(bool &x) : f(x) { f = true; }
    ~InBatch() { f = false; }
  } in_batch_guard(p.in_batch);
  const uint64_t kChunk = 1ull << 30;
  for (uint64_t off = 0; off < n; off += kChunk) {
    const uint64_t m = std::min(kChunk, n - off);
    // a speculative epoch has a fixed cost (rollback snapshot of the whole array, stamp resets: ~180 us at 16 M slots);
    // a handful of updates — the single-update API above all — is cheaper through the strict prefix rounds, which need
    // neither (same result: both are exact)
    const bool small = m <= p.small_batch;
    int rc;
    if (p.v.g.narrow == 0u) {
      rc = recheck_ranges();  // (cheap next to one-update-per-round execution)
      if (rc != PPCSR_OK) return rc;
    }
    if (p.v.g.narrow == 0u) {
      // The reference's add_node-after-doubling path has left a structure whose vertex ranges overlap (see
      // run_exclusive): footprints computed from sorted, disjoint ranges no longer describe what an update touches, so
      // nothing may run side by side — one update per round, i.e. plain sequential execution of the exact kernels.
      const uint32_t mh = p.max_horizon, mi = p.min_horizon, ih = p.init_horizon;
      p.max_horizon = p.min_horizon = p.init_horizon = 1;
      rc = run_rounds(d_ops + off, m);
      p.max_horizon = mh;
      p.min_horizon = mi;
      p.init_horizon = ih;
    } else {
      rc = (p.mode == 1 && !small) ? run_speculative(d_ops + off, m) : run_rounds(d_ops + off, m);
    }
    if (rc != PPCSR_OK) return rc;
  }
  p.timer.stop(p.stream);
  p.st.last_batch_ms = p.timer.ms();
  p.st.ops_applied += n;
  return inplace_fault_check();
}

int Engine::run_rounds(const Op *d_ops, uint64_t n) {
  Impl &p = *p_;
  if (p.round > 0xFFFF0000u) GCHK(reset_tags(p));  // reservation tags would wrap
  GCHK(ensure_plans(p));
  uint64_t cur = 0;
  uint32_t hor = (uint32_t)std::min<uint64_t>(p.init_horizon, n);
  while (cur < n) {
    // (re)arm the control block for the next round's parity
    const uint32_t par = (p.round + 1) & 1u;
    Control &c = *p.h_ctl;
    c.base[par] = (uint32_t)cur;
    c.base[par ^ 1u] = (uint32_t)cur;
    c.horizon[par] = (uint32_t)std::min<uint64_t>(hor, n - cur);
    c.horizon[par ^ 1u] = 0;
    c.failmin[0] = c.failmin[1] = kMax;
    c.n_ops = (uint32_t)n;
    c.excl = 0;
    c.error = 0;
    c.max_horizon = p.max_horizon;
    c.rounds = c.committed = c.planned = 0;
    GCHK(gpu::h2d(p.d_ctl, p.h_ctl, sizeof(Control), p.stream));
    bool need_excl = false;
    while (cur < n && !need_excl) {
      RoundArgs a;
      a.v = p.v;
      a.ops = d_ops;
      a.plans = p.d_plans;
      a.ctl = p.d_ctl;
      a.stats = p.d_stats;
      a.min_horizon = p.min_horizon;
      const uint32_t blocks = (p.max_horizon + 3) / 4;
      if (p.profile && p.events.size() < 4ull * p.rounds_per_sync) {
        const size_t old = p.events.size();
        p.events.resize(4ull * p.rounds_per_sync);
        for (size_t i = old; i < p.events.size(); i++) GCHK(p.events[i].init());
      }
      // rounds in this chunk: at least one per `hor` pending updates (every strict round commits at least one update and
      // at most the horizon), capped by rounds_per_sync; short batches — the single-update API — get 1-2 rounds, not 32
      uint32_t chunk_rounds = p.rounds_per_sync;
      {
        const uint64_t pending = n - cur;
        const uint64_t need = (pending + std::max<uint32_t>(hor, 1u) - 1) / std::max<uint32_t>(hor, 1u) + 1;
        if (need < chunk_rounds) chunk_rounds = (uint32_t)need;
      }
      for (uint32_t r = 0; r < chunk_rounds; r++) {
        a.round = ++p.round;
        if (p.profile) p.events[4 * r + 0].record(p.stream);
        GPU_LAUNCH(p.stream, k_plan, blocks, 256, a);
        if (p.profile) p.events[4 * r + 1].record(p.stream);
        GPU_LAUNCH(p.stream, k_check, blocks, 256, a);
        if (p.profile) p.events[4 * r + 2].record(p.stream);
        GPU_LAUNCH(p.stream, k_apply, blocks, 256, a);
        if (p.profile) p.events[4 * r + 3].record(p.stream);
      }
      GCHK(gpu::d2h(p.h_ctl, p.d_ctl, sizeof(Control), p.stream));
      GCHK(gpu::sync(p.stream));
      GCHK(gpu::last_error());
      if (p.profile) {
        for (uint32_t r = 0; r < chunk_rounds; r++) {
          p.st.prof_plan_ms += gpu::Event::elapsed_ms(p.events[4 * r + 0], p.events[4 * r + 1]);
          p.st.prof_check_ms += gpu::Event::elapsed_ms(p.events[4 * r + 1], p.events[4 * r + 2]);
          p.st.prof_apply_ms += gpu::Event::elapsed_ms(p.events[4 * r + 2], p.events[4 * r + 3]);
          p.st.prof_launches += 1;
        }
      }
      p.st.round_syncs++;
      if (c.error) return fail(PPCSR_EINTERNAL, "device-side error " + std::to_string(c.error));
      const uint32_t npar = (p.round + 1) & 1u;
      cur = c.base[npar];
      hor = std::max<uint32_t>(c.horizon[npar], p.min_horizon);
      need_excl = c.excl != 0;
      p.st.rounds += c.rounds;
      p.st.committed += c.committed;
      p.st.planned += c.planned;
      c.rounds = c.committed = c.planned = 0;
      if (!need_excl && cur < n) {
        // counters were consumed on the host; clear them on the device without touching base/horizon
        GCHK(gpu::dset(&p.d_ctl->rounds, 0, 3 * sizeof(unsigned long long), p.stream));
      }
    }
    if (need_excl && cur < n) {
      GCHK(gpu::d2h(p.h_op1, d_ops + cur, sizeof(Op), p.stream));
      GCHK(gpu::sync(p.stream));
      int rc = run_exclusive(*p.h_op1, 0);
      if (rc != PPCSR_OK) return rc;
      cur += 1;
    }
  }
  return PPCSR_OK;
}

static int snap_commit(Engine::Impl &p, Engine::Impl::Snap &sn);
static int snap_rollback(Engine::Impl &p, Engine::Impl::Snap &sn, Engine::Impl::Snap &other);

// Speculative rounds (pma_kernels.h, second half): epochs of at most `epoch_ops` updates, each with a rollback
// snapshot; a validation failure replays the epoch with the strict prefix rounds, an exclusive update ends the epoch.
int Engine::run_speculative(const Op *d_ops, uint64_t n) {
  Impl &p = *p_;
  GCHK(ensure_plans(p));
  // per-slot and carry arrays
  if (p.hslot_cap < p.opt_horizon) {
    if (p.d_status) GPU_DFREE(p.d_status);
    p.d_status = nullptr;
    p.hslot_cap = 0;
    GCHK(gpu::dmalloc((void **)&p.d_status, ((uint64_t)p.opt_horizon + 64) * sizeof(uint32_t)));  // (+64: padded to the launch grids)
    if (p.d_vdbg) GPU_DFREE(p.d_vdbg);
    GCHK(gpu::dmalloc((void **)&p.d_vdbg, ((uint64_t)p.opt_horizon + 64) * 4 * sizeof(uint32_t)));
    p.hslot_cap = p.opt_horizon;
  }
  // carry lists: by round parity, kStripes sub-lists each (a round's deferred updates can all come from workgroups of one XCD)
  const uint64_t carry_need = (uint64_t)p.opt_horizon + 8;
  if (p.carry_cap < carry_need) {
    if (p.d_carry0) GPU_DFREE(p.d_carry0);
    if (p.d_carry1) GPU_DFREE(p.d_carry1);
    p.d_carry0 = p.d_carry1 = nullptr;
    p.carry_cap = 0;
    GCHK(gpu::dmalloc((void **)&p.d_carry0, kStripes * carry_need * sizeof(uint32_t)));
    GCHK(gpu::dmalloc((void **)&p.d_carry1, kStripes * carry_need * sizeof(uint32_t)));
    p.carry_cap = carry_need;
  }
  // big-window path: windows of up to kBigLeaves leaves (and big_window slots) stay inside the round
  uint32_t bigw = p.big_window;
  while (bigw > kBigWindow && (bigw >> p.v.g.sh) > dev::kBigLeaves) bigw >>= 1;
  const bool use_big = bigw > p.big_min;
  if (use_big) {
    if (!p.d_jobs) GCHK(gpu::dmalloc((void **)&p.d_jobs, kBigJobs * sizeof(dev::BigJob)));
    const uint64_t need = (uint64_t)p.big_grid * bigw;
    if (p.bigscratch_cap < need) {
      if (p.d_bigscratch) GPU_DFREE(p.d_bigscratch);
      p.d_bigscratch = nullptr;
      p.bigscratch_cap = 0;
      GCHK(gpu::dmalloc((void **)&p.d_bigscratch, need * sizeof(Edge)));
      p.bigscratch_cap = need;
    }
  }
  if (p.diag >= 2) {
    if (p.dg_cap < n) {
      if (p.d_dg) GPU_DFREE(p.d_dg);
      p.d_dg = nullptr;
      p.dg_cap = 0;
      GCHK(gpu::dmalloc((void **)&p.d_dg, n * 4 * sizeof(uint32_t)));
      p.dg_cap = n;
    }
    GCHK(gpu::dset(p.d_dg, 0, n * 4 * sizeof(uint32_t), p.stream));
  }
  p.stamps_clean = false;  // (stream indices start over with every batch)
  uint64_t e0 = 0;
  uint64_t forced_e1 = 0;  // after a rollback: end the retried epoch right after the update that failed validation
  int retries = 0;
  const uint32_t kEpochShort = p.epoch_short;
  if (p.cur_epoch == 0 || p.cur_epoch > p.epoch_ops) p.cur_epoch = p.epoch_ops;
  while (e0 < n) {
    uint64_t e1 = std::min<uint64_t>(e0 + p.cur_epoch, n);
    if (forced_e1 > e0 && forced_e1 < e1) e1 = forced_e1;
    if (p.round > 0xFFFF0000u) GCHK(reset_tags(p));
    // rollback point of the epoch: the epoch snapshot catches up with what the previous epoch wrote (dirty tags) — a full
    // copy only the first time and after the array was replaced
    GCHK(snap_commit(p, p.esnap));
    GCHK(gpu::d2d(p.d_stats_snap, p.d_stats, kStatShards * sizeof(StatShard), p.stream));
    if (!p.stamps_clean) {
      // validation stamps hold 1 + the stream index of the latest committed toucher: what earlier epochs of this batch left
      // is smaller than anything this epoch compares with, so they are cleared at the start of a batch and after a
      // rollback only (the rolled-back commits must not look like later updates to the retried ones)
      const uint64_t leaves = p.v.g.N >> p.v.g.sh;
      GCHK(gpu::dset(p.d_wstamp, 0, leaves * sizeof(uint32_t), p.stream));
      GCHK(gpu::dset(p.d_rstamp, 0, leaves * sizeof(uint32_t), p.stream));
      GCHK(gpu::dset(p.d_vws, 0, (p.n_cap + 1) * sizeof(uint32_t), p.stream));
      GCHK(gpu::dset(p.d_vrs, 0, (p.n_cap + 1) * sizeof(uint32_t), p.stream));
      p.stamps_clean = true;
    }
    OptCtl &c = *p.h_octl;
    memset(&c, 0, sizeof(c));
    // the epoch's first round is p.round + 1: the words it reads as "the previous round" describe an empty round that ended at e0
    const uint32_t pp0 = p.round & 1u;
    c.hor[pp0] = 0;
    c.used[pp0] = 0;
    c.nf[pp0] = (uint32_t)e0;
    c.adaptive = p.adaptive;
    c.cur_h[pp0] = p.adaptive ? std::min(p.start_horizon, p.opt_horizon) : p.opt_horizon;
    c.book_round = p.round;  // (nothing to add to the counters for it)
    for (int t = 0; t < 3; t++) {
      c.minkept[t] = kMax;
      c.gbar[t] = c.sbar[t] = ~0ull;
    }
    c.e1 = (uint32_t)e1;
    c.max_horizon = p.opt_horizon;
    c.width_cap = p.opt_horizon;
    c.resident = p.resident_waves;
    c.viol_idx = kMax;
    c.skip_idx = kMax;
    c.skip_round = 0;
    GCHK(gpu::h2d(p.d_octl, p.h_octl, sizeof(OptCtl), p.stream));
    int rs = 0;
    if (p.region_eff < p.region_slots) p.region_eff = p.region_slots;
    while ((p.region_eff >> rs) > (uint32_t)p.v.g.logN) rs++;
    bool epoch_open = true;
    uint32_t cur_width = c.cur_h[pp0];  // the adapted width the device last reported
    uint64_t carry_now = 0;            // deferred updates waiting in the carry list
    unsigned long long jobs_seen = 0, rounds_seen = 0, planned_seen = 0, long_seen = 0;  // OptCtl counters at the last look
    // tail sizing of the round chunks: updates still pending and updates committed per round (measured on the last chunk)
    uint64_t chunk_pending = e1 - e0;
    double chunk_cpr = 0.9 * (double)std::min<uint64_t>(cur_width, e1 - e0);
    unsigned long long prev_rounds = 0, prev_committed = 0;
    uint32_t excl_cooldown = 0;  // chunks to keep short after an exclusive update (the launches behind it in its chunk are wasted)
    while (epoch_open) {
      OptArgs a;
      a.v = p.v;
      a.v.big_window = use_big ? bigw : kBigWindow;
      a.big_min = p.big_min;
      a.jobs = use_big ? p.d_jobs : (dev::BigJob *)nullptr;
      a.bigscratch = p.d_bigscratch;
      a.bigscratch_stride = bigw;
      a.ops = d_ops;
      a.plans = p.d_plans;
      a.status = p.d_status;
      a.vdbg = p.d_vdbg;
      a.carry0 = p.d_carry0;
      a.carry1 = p.d_carry1;
      a.carry_cap = (uint32_t)p.carry_cap;
      a.ctl = p.d_octl;
      a.stats = p.d_stats;
      a.regfail = p.d_regfail;
      a.pfail = p.d_pfail;
      a.wstamp = p.d_wstamp;
      a.rstamp = p.d_rstamp;
      a.vws = p.d_vws;
      a.vrs = p.d_vrs;
      a.regshift = rs;
      a.diag = p.diag;
      a.defer_barrier = p.defer_barrier;
      // the round kernels come in two instantiations: without / with the diagnostics compiled in
      const bool extras = p.diag != 0 || p.dbg_repeat != 0;
      a.dbg = p.dbg_repeat;
      a.dg = p.diag >= 2 ? p.d_dg : (uint32_t *)nullptr;
      a.soft_barrier = p.soft_barrier ? p.soft_barrier : a.v.big_window / 2u;
      // while the rare-rollback rule keeps regions wide, overtakers are kept off a growing window by the region rule itself and
      // the soft barrier only costs commits (a config #4 partition alone: 156 -> 96 rounds, 19.6 -> 13.9 ms, still no rollback)
      if (!p.soft_barrier && p.region_rare_span && p.region_eff >= p.region_rare) a.soft_barrier = 0x40000000u;
      // grid sized for the horizon the device last reported (it can only shrink within a chunk when fresh
      // updates run out; it never exceeds opt_horizon)
      // grid: wide enough for the adapted width to grow during the chunk (x1.25 per full-width round), narrow at the tail
      uint32_t gh = p.opt_horizon;
      {
        const uint64_t want = std::max<uint64_t>(1024, 4ull * (cur_width ? cur_width : p.opt_horizon));
        if (want < gh) gh = (uint32_t)want;
        const uint64_t pend = (chunk_pending + 255) & ~255ull;
        if (pend < gh) gh = (uint32_t)std::max<uint64_t>(pend, 256);
        // (a round plans the whole carry list: the grid is never narrower than it)
        const uint64_t need = (carry_now + 255) & ~255ull;
        if (gh < need) gh = (uint32_t)std::min<uint64_t>(need, p.opt_horizon);
      }
      const uint32_t blocks = (gh + 3) / 4;
      if (gh != c.max_horizon) {  // the device must never choose a horizon larger than the launched grid
        GCHK(gpu::h2d(&p.d_octl->max_horizon, &gh, sizeof(uint32_t), p.stream));
        c.max_horizon = gh;
      }
      // rounds in this chunk: as many as rounds_per_sync while that many are needed, fewer at the tail of the epoch so
      // that the stream is not padded with launches that find `done` set (each still costs ~2 us of dispatch)
      uint32_t rounds = p.rounds_per_sync;
      {
        const double per_round = std::max(1.0, 0.85 * chunk_cpr);
        const double est = (double)chunk_pending / per_round + 2.0;
        if (est < (double)rounds) rounds = (uint32_t)std::max(2.0, est);
      }
      if (excl_cooldown) {
        rounds = std::min<uint32_t>(rounds, 8u);
        excl_cooldown--;
      }
      if (p.profile && p.events.size() < 5ull * rounds) {
        const size_t oldn = p.events.size();
        p.events.resize(5ull * rounds);
        for (size_t i = oldn; i < p.events.size(); i++) GCHK(p.events[i].init());
      }
      for (uint32_t r = 0; r < rounds; r++) {
        a.round = ++p.round;
        if (p.profile) p.events[5 * r + 0].record(p.stream);
        if (extras) GPU_LAUNCH(p.stream, o_plan_x, blocks, 256, a); else GPU_LAUNCH(p.stream, o_plan, blocks, 256, a);
        if (p.profile) p.events[5 * r + 1].record(p.stream);
        // (o_check: a LANE per update, one wave per workgroup; the diagnostics build keeps a wave per update)
        if (extras) GPU_LAUNCH(p.stream, o_check_x, blocks, 256, a);
        else if (p.check_force >= 0 ? p.check_force == 1 : p.check_lanes) GPU_LAUNCH(p.stream, o_check, (gh + kCkThreads - 1) / kCkThreads, kCkThreads, a);
        else GPU_LAUNCH(p.stream, o_check_w, blocks, 256, a);
        if (p.profile) p.events[5 * r + 2].record(p.stream);
        if (extras) GPU_LAUNCH(p.stream, o_apply_x, blocks, 256, a); else GPU_LAUNCH(p.stream, o_apply, blocks, 256, a);
        if (p.profile) p.events[5 * r + 3].record(p.stream);
        if (use_big && p.big_on) GPU_LAUNCH(p.stream, o_big, p.big_grid, 1024, a);  // the round's queued big-window rebalances (profile: the 'compact' column)
        if (p.profile) p.events[5 * r + 4].record(p.stream);
      }
      {  // the last round's outcome, recorded for the host (the next round's o_plan will find the same)
        a.round = p.round + 1;
        GPU_LAUNCH(p.stream, o_settle, 1, 64, a);
      }
      GCHK(gpu::d2h(p.h_octl, p.d_octl, sizeof(OptCtl), p.stream));
      GCHK(gpu::sync(p.stream));
      GCHK(gpu::last_error());
      if (p.profile) {
        for (uint32_t r = 0; r < rounds; r++) {
          p.st.prof_plan_ms += gpu::Event::elapsed_ms(p.events[5 * r + 0], p.events[5 * r + 1]);
          p.st.prof_check_ms += gpu::Event::elapsed_ms(p.events[5 * r + 1], p.events[5 * r + 2]);
          p.st.prof_apply_ms += gpu::Event::elapsed_ms(p.events[5 * r + 2], p.events[5 * r + 3]);
          if (use_big && p.big_on) p.st.prof_compact_ms += gpu::Event::elapsed_ms(p.events[5 * r + 3], p.events[5 * r + 4]);  // (o_big, when it is in)
          p.st.prof_launches += 1;
        }
      }
      p.st.round_syncs++;
      if (c.error) return fail(PPCSR_EINTERNAL, "device-side error " + std::to_string(c.error));
      {  // which o_check for the next chunk: see Impl::check_lanes.  The lane kernel counts what it had to leave to the wave
         // (each such update costs its wave a serial ~5 us): above ~1 in 64 a wave per update from the start is faster.  From the
         // wave kernel the way back is a trial chunk every so often.
        const unsigned long long dp = c.planned - planned_seen, dl = c.long_checks - long_seen;
        planned_seen = c.planned;
        long_seen = c.long_checks;
        if (p.check_lanes) {
          if (dp >= 1024 && dl * 64ull > dp) {
            p.check_lanes = false;
            p.check_wave_chunks = 0;
          }
        } else if (++p.check_wave_chunks >= 24u) {
          p.check_lanes = true;
        }
      }
      if (use_big) {  // keep / drop the o_big launch: see Impl::big_on.  Kept while a stream queues a window every few rounds (an
                      // empty launch costs ~4 us, a round that finds none where it needs one costs the rest of its chunk)
        const unsigned long long dj = c.jobs_total - jobs_seen, dr = c.rounds - rounds_seen;
        jobs_seen = c.jobs_total;
        rounds_seen = c.rounds;
        if (dr) p.big_rate = 0.5 * p.big_rate + 0.5 * (double)dj / (double)dr;  // queued windows per round, running mean
        p.big_on = p.big_rate >= 1.0 / 12.0;
      }
      if (c.need_big && !c.violation) {
        // the last round that ran (c.book_round) queued windows and no o_big followed: run it now, forget the launches that
        // found the flag, go on from the next round with the launch back in
        const uint32_t R = c.book_round;
        a.round = R;
        p.h_octl->need_big = 0;
        GCHK(gpu::h2d(&p.d_octl->need_big, &p.h_octl->need_big, sizeof(uint32_t), p.stream));  // (first: o_big stands back while a flag is up)
        GPU_LAUNCH(p.stream, o_big, p.big_grid, 1024, a);
        GCHK(gpu::sync(p.stream));
        p.round = R;
        const uint32_t np2 = (R + 1u) & 1u;
        cur_width = c.cur_h[np2];
        carry_now = c.used[np2];
        chunk_pending = (uint64_t)c.used[np2] + (uint64_t)(c.e1 - c.nf[np2]);
        continue;
      }
      if (p.diag && (c.violation || c.excl || c.done))
        fprintf(stderr, "[ppcsr diag] epoch [%llu,%llu) %s after %llu rounds: committed %llu planned %llu | not committed because: excl-kind %llu, "
                "behind-barrier %llu, dup %llu, W-W %llu, W-after-R %llu, R-after-W %llu, sentinel-read %llu, sentinel-move %llu, region %llu, "
                "growth-zone %llu, stamp %llu\n",
                (unsigned long long)e0, (unsigned long long)e1, c.violation ? "ROLLBACK" : (c.excl ? "exclusive" : "done"), c.rounds, c.committed,
                c.planned, c.why[0], c.why[1], c.why[2], c.why[3], c.why[4], c.why[5], c.why[6], c.why[7], c.why[8], c.why[9], c.why[10]);
      if (p.diag >= 2 && (c.violation || c.done) && getenv("PPCSR_DIAG_DUMP")) {
        const uint64_t cnt = e1 - e0;
        std::vector<uint32_t> tr(cnt * 4);
        std::vector<Op> hops(cnt);
        GCHK(gpu::d2h(tr.data(), p.d_dg + 4 * e0, cnt * 4 * sizeof(uint32_t), p.stream));
        GCHK(gpu::d2h(hops.data(), d_ops + e0, cnt * sizeof(Op), p.stream));
        GCHK(gpu::sync(p.stream));
        if (FILE *f = fopen(getenv("PPCSR_DIAG_DUMP"), "ab")) {
          const uint64_t hdr[4] = {e0, e1, c.rounds, c.violation ? 1ull : 0ull};
          fwrite(hdr, sizeof(hdr), 1, f);
          fwrite(tr.data(), sizeof(uint32_t), tr.size(), f);
          fwrite(hops.data(), sizeof(Op), hops.size(), f);
          fclose(f);
        }
        GCHK(gpu::dset(p.d_dg + 4 * e0, 0, cnt * 4 * sizeof(uint32_t), p.stream));  // (a retried epoch starts its trace over)
      }
      // (after o_settle: what the next round would plan, by ITS parity)
      const uint32_t npar = (c.book_round + 1u) & 1u;
      // An exclusive update runs now, alone, in the middle of the epoch: it is the lowest pending update, so it sees exactly
      // the state sequential execution gives it unless a LATER update was committed earlier on something it reads or
      // writes — validated with the stamps like every other update (k_exclusive, XValid).  Only a resize (double_list /
      // half_list rewrite the whole array and its geometry) keeps the blanket rule "nothing later may have been committed"
      // and ends the epoch; everything else continues it: the update's slot commits as nothing in the next round (K_SKIP).
      bool excl_resized = false;
      if (c.excl && !c.violation) {
        const uint64_t g = c.excl_idx;
        bool viol = false;
        int rc = run_exclusive(Op{0, 0, 0}, 0, d_ops, (uint32_t)g, &viol, &excl_resized, c.excl_later != 0);
        if (rc != PPCSR_OK) return rc;
        if (viol) {
          c.violation = 1;
          c.viol_idx = c.excl_idx;
        }
      }
      if (c.violation) {
        // a later update was committed before an earlier one whose footprint then reached it: roll the epoch back
        // and retry it cut right after that update (nothing can overtake the last update of an epoch); after
        // repeated failures replay the epoch with the strict prefix rounds
        p.st.rollbacks++;
        if (getenv("PPCSR_TRACE_EPOCH"))
          fprintf(stderr, "[ppcsr] rollback: epoch [%llu,%llu) viol_idx=%u excl=%u later=%u after %llu rounds; kind=%u leaf=%u stamp=%u what=%u wleaf=[%u,%u] index=%u nr=%u\n",
                  (unsigned long long)e0, (unsigned long long)e1, c.viol_idx, c.excl, c.excl_later, c.rounds, c.viol_info[0],
                  c.viol_info[1], c.viol_info[2], c.viol_info[3], c.viol_info[4], c.viol_info[5], c.viol_info[6], c.viol_info[7]);
        GCHK(snap_rollback(p, p.esnap, p.snap));
        p.stamps_clean = false;
        GCHK(gpu::d2d(p.d_stats, p.d_stats_snap, kStatShards * sizeof(StatShard), p.stream));
        p.st.wasted_rounds += c.rounds;  // (kept apart: `rounds` / `committed` / `planned` describe committed work only)
        uint32_t short_eff = kEpochShort;
        p.grow_eff = p.epoch_grow_after;
        if (p.epoch_adapt) {
          const double D = (double)std::min<uint64_t>(p.since_rollback, 64ull * kEpochShort);
          // (running mean with a fast attack downwards: a stream that starts to roll back often is recognised at once)
          p.rb_dist = p.rb_dist < 0 ? 64.0 * kEpochShort : (D < p.rb_dist ? 0.25 * p.rb_dist + 0.75 * D : 0.5 * p.rb_dist + 0.5 * D);
          p.since_rollback = 0;
          uint32_t q = 2048;
          while (2ull * q <= (uint64_t)(p.rb_dist / (double)p.epoch_adapt) && 2ull * q <= kEpochShort) q *= 2;
          short_eff = std::min<uint32_t>(kEpochShort, q);
          if (short_eff < kEpochShort) p.grow_eff = std::max<uint32_t>(p.epoch_grow_after, 4u);
        }
        p.cur_epoch = std::min<uint32_t>(p.cur_epoch, std::min<uint32_t>(p.epoch_ops, short_eff));
        p.epoch_clean = 0;
        {
          const bool rare = p.epoch_adapt && p.region_rare && p.rb_dist >= (double)p.region_rare_dist && p.cpr_mean >= (double)p.region_rare_cpr;
          p.region_eff = std::max(p.region_slots, rare ? std::max(p.region_wide, p.region_rare) : p.region_wide);
          p.region_rare_span = rare ? (uint64_t)(p.region_rare_calm * p.rb_dist) : 0;
        }
        p.region_clean = 0;
        const uint64_t cut = (uint64_t)c.viol_idx + 1;
        if (retries < 3 && c.viol_idx != kMax && cut > e0 && cut < e1) {
          retries++;
          forced_e1 = cut;
        } else {
          int rc = run_rounds(d_ops + e0, e1 - e0);
          if (rc != PPCSR_OK) return rc;
          e0 = e1;
          retries = 0;
          forced_e1 = 0;
        }
        epoch_open = false;
      } else if (c.excl && excl_resized) {
        p.st.rounds += c.rounds;
        p.st.committed += c.committed + 1;
        p.st.planned += c.planned;
        e0 = (uint64_t)c.excl_idx + 1;
        retries = 0;
        epoch_open = false;
      } else if (c.excl) {
        // the epoch goes on.  The launches queued behind the update's discovery did nothing; the next round takes the number
        // after the last round that ran (c.book_round), finds that round's words as they were, minus the barrier the
        // exclusive update raised (or it would be discovered again), and commits the update's slot as nothing (K_SKIP)
        const uint32_t R = c.book_round;
        p.h_octl->excl = 0;
        p.h_octl->skip_idx = c.excl_idx;
        p.h_octl->skip_round = R + 1u;
        p.h_octl->gbar[R % 3u] = ~0ull;
        GCHK(gpu::h2d(&p.d_octl->excl, &p.h_octl->excl, sizeof(uint32_t), p.stream));
        GCHK(gpu::h2d(&p.d_octl->skip_idx, &p.h_octl->skip_idx, 2 * sizeof(uint32_t), p.stream));
        GCHK(gpu::h2d(&p.d_octl->gbar[R % 3u], &p.h_octl->gbar[R % 3u], sizeof(unsigned long long), p.stream));
        GCHK(gpu::sync(p.stream));  // (h_octl is the landing buffer of the next chunk's control block)
        p.round = R;
        cur_width = c.cur_h[npar];
        carry_now = c.used[npar];
        chunk_pending = (uint64_t)c.used[npar] + (uint64_t)(c.e1 - c.nf[npar]);
        excl_cooldown = 4;
      } else if (c.done) {
        if (getenv("PPCSR_TRACE_EPOCH"))
          fprintf(stderr, "[ppcsr] epoch [%llu,%llu) rounds=%llu planned=%llu committed=%llu\n", (unsigned long long)e0, (unsigned long long)e1, c.rounds,
                  c.planned, c.committed);
        p.st.rounds += c.rounds;
        p.st.committed += c.committed;
        p.st.planned += c.planned;
        p.since_rollback += e1 - e0;
        if (c.rounds) {
          const double cpr = (double)c.committed / (double)c.rounds;
          p.cpr_mean = p.cpr_mean < 0 ? cpr : 0.75 * p.cpr_mean + 0.25 * cpr;
        }
        e0 = e1;
        retries = 0;
        epoch_open = false;
        if (p.region_eff != p.region_slots) {
          ++p.region_clean;
          if (p.region_rare_span ? p.since_rollback >= p.region_rare_span : p.region_clean >= p.region_calm) {
            p.region_eff = p.region_slots;
            p.region_rare_span = 0;
          }
        }
        if (++p.epoch_clean >= std::max(p.grow_eff, p.epoch_grow_after)) {
          p.cur_epoch = (uint32_t)std::min<uint64_t>(p.epoch_ops, 2ull * p.cur_epoch);
          p.epoch_clean = 0;
        }
        if (c.cur_h[npar]) p.start_horizon = c.cur_h[npar];  // keep the adapted width for the next epoch
      } else {
        cur_width = c.cur_h[npar];
        carry_now = c.used[npar];
        chunk_pending = (uint64_t)c.used[npar] + (uint64_t)(c.e1 - c.nf[npar]);
        if (c.rounds > prev_rounds) chunk_cpr = (double)(c.committed - prev_committed) / (double)(c.rounds - prev_rounds);
        prev_rounds = c.rounds;
        prev_committed = c.committed;
      }
    }
  }
  return PPCSR_OK;
}

int Engine::run_exclusive(Op op, uint32_t flags, const Op *d_ops, uint32_t spec_index, bool *violation, bool *resized, bool later_committed) {
  Impl &p = *p_;
  if (violation) *violation = false;
  if (resized) *resized = false;
  const bool spec = spec_index != kMax;
  bool counted = false;  // exclusive_ops counts executed updates: not the attempts, not a run that is rolled back
  auto count_once = [&]() {
    if (!counted) p.st.exclusive_ops++;
    counted = true;
  };
  for (int attempt = 0; attempt < 8; attempt++) {
    XValid xv;
    xv.wstamp = p.d_wstamp;
    xv.rstamp = p.d_rstamp;
    xv.vws = p.d_vws;
    xv.me1 = (spec && attempt == 0) ? spec_index + 1u : 0u;  // (a retry follows a doubling: whole-array rule, see the caller)
    const auto tx0 = std::chrono::steady_clock::now();
    GPU_LAUNCH(p.stream, k_exclusive, 1, 64, p.v, op, d_ops, spec ? spec_index : kMax, flags, p.d_xout, p.d_stats, p.excl_in_wave, xv);
    GCHK(gpu::d2h(p.h_xout, p.d_xout, sizeof(ExclOut), p.stream));
    GCHK(gpu::sync(p.stream));
    GCHK(gpu::last_error());
    const ExclOut x = *p.h_xout;
    if (getenv("PPCSR_TRACE_EXCL"))
      fprintf(stderr, "[excl] op (%u,%u,%u) flags %u -> result %u window (%u,%u) found %u, %.0f us\n", op.src, op.dst, op.op, flags, x.result, x.wstart, x.wlen, x.found,
              std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tx0).count());
    switch (x.result) {
      case X_DONE:
        count_once();
        return PPCSR_OK;
      case X_VIOLATION:
        if (violation) *violation = true;
        return PPCSR_OK;
      case X_NEED_DOUBLE:
      case X_NEED_HALF:
        // double_list / half_list rewrite the whole array and its geometry: inside a speculative epoch they are only
        // serialisable when nothing later has been committed yet (otherwise: roll back, the retry ends the epoch here)
        if (spec && later_committed) {
          if (violation) *violation = true;
          return PPCSR_OK;
        }
        if (resized) *resized = true;
        count_once();
        return resize(x.result == X_NEED_DOUBLE ? p.v.g.N * 2 : p.v.g.N / 2);
      case X_NEED_REDIST:
        count_once();
        // (inside a batch the in-place rebalance's sticky fault flag is read once, at the end of apply_batch_device; a
        // single-update caller — add_node, the strict rounds of a one-update batch — has it checked right here)
        return big_redistribute(x.wstart, x.wlen, !spec && !p.in_batch);
      case X_DOUBLE_THEN_RETRY: {
        if (spec && later_committed) {
          if (violation) *violation = true;
          return PPCSR_OK;
        }
        if (resized) *resized = true;
        int rc = resize(p.v.g.N * 2);
        if (rc != PPCSR_OK) return rc;
        count_once();
        flags |= XF_FORCE_NOINFO | XF_SKIP_COUNT;
        if (flags & XF_ADD_NODE) {
          // add_node found the end of the array occupied: the reference doubles and then re-searches the new sentinel's
          // place through a node record the doubling could not update (PCSR.cpp:533-540, 681-703).  The sentinel can land
          // in the middle of another vertex's range; the reference then searches unsorted ranges, which only the literal
          // walk reproduces — the 64-ary narrowing is switched off for good.
          flags |= XF_RESEARCH;
          p.v.g.narrow = 0u;
          p.st.narrow_lost++;
        }
        break;
      }
      case X_UNSUPPORTED: return fail(PPCSR_EUNSUPPORTED, error_string(PPCSR_EUNSUPPORTED));
      case X_WINDOW_BEYOND_ARRAY:
        return fail(PPCSR_EUNSUPPORTED, "rebalance window exceeds a one-leaf array (the reference reads past the end of its array here: undefined behaviour)");
      default: return fail(PPCSR_EINTERNAL, "bad exclusive result");
    }
  }
  return fail(PPCSR_EINTERNAL, "exclusive executor did not converge");
}

// add_node after a doubling can drop the new sentinel into another vertex's range (the reference's own behaviour); from
// then on only sequential execution follows the reference.  But very often the sentinel lands in a sane place — or a later
// resize sorts things out — and the structure is as regular as ever: check, and leave the sequential regime again.
int Engine::recheck_ranges() {
  Impl &p = *p_;
  if (p.v.g.narrow != 0u) return PPCSR_OK;
  const unsigned long long zero = 0;
  *p.h_total = zero;
  GCHK(gpu::h2d(p.d_total, p.h_total, sizeof(unsigned long long), p.stream));
  GPU_LAUNCH(p.stream, k_check_ranges, grid_for(std::max<uint32_t>(p.v.g.n, 1u), 4, 16384), 256, p.v, p.d_total);
  GCHK(gpu::d2h(p.h_total, p.d_total, sizeof(unsigned long long), p.stream));
  GCHK(gpu::sync(p.stream));
  GCHK(gpu::last_error());
  if (*p.h_total == 0) p.v.g.narrow = 1u;
  return PPCSR_OK;
}

int Engine::ensure_scratch(uint64_t nleaves) {
  Impl &p = *p_;
  if (nleaves > p.rank_cap) {
    if (p.d_rank) GPU_DFREE(p.d_rank);
    p.d_rank = nullptr;
    p.rank_cap = 0;
    GCHK(gpu::dmalloc((void **)&p.d_rank, nleaves * sizeof(uint32_t)));
    p.rank_cap = nleaves;
  }
  return ensure_tiles((nleaves + kScanTile - 1) / kScanTile);
}

int Engine::ensure_tiles(uint64_t ntiles) {
  Impl &p = *p_;
  if (ntiles > p.tiles_cap) {
    if (p.d_tiles) GPU_DFREE(p.d_tiles);
    p.d_tiles = nullptr;
    p.tiles_cap = 0;
    GCHK(gpu::dmalloc((void **)&p.d_tiles, ntiles * sizeof(uint32_t)));
    p.tiles_cap = ntiles;
  }
  return PPCSR_OK;
}

// exclusive prefix sum of d_cnt[0..nleaves) into d_rank_, grand total into d_total_
int Engine::rank_scan(const uint32_t *d_cnt, uint64_t nleaves, bool table, uint64_t tb_index, uint64_t tb_len) {
  Impl &p = *p_;
  int rc = ensure_scratch(nleaves);
  if (rc != PPCSR_OK) return rc;
  const uint64_t ntiles = (nleaves + kScanTile - 1) / kScanTile;
  GPU_LAUNCH(p.stream, k_scan_tiles, ntiles, 256, d_cnt, nleaves, p.d_tiles);
  GPU_LAUNCH(p.stream, k_scan_tilesums, 1, kTileSumThreads, p.d_tiles, ntiles, p.d_total, table ? p.d_table : (ChainTable *)nullptr, tb_index, tb_len, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u, 0u);
  GPU_LAUNCH(p.stream, k_scan_apply, ntiles, 256, d_cnt, nleaves, (const uint32_t *)p.d_tiles, p.d_rank);
  return PPCSR_OK;
}

// three-launch rebalance pipeline (scatter_variant 2): tile sums (+ zeroing of the destination leaf counts), tile scan +
// position table, scatter with the in-tile leaf scan done in LDS.  The source leaf counts must stay intact while the
// scatter runs, so an in-place window (source counts == destination counts) is served from a copy of its counts.
int Engine::rebalance_fused(const View &nv, const Edge *src_items, uint64_t src_lo, uint64_t src_len, int src_sh,
                            uint32_t *src_cnt, bool inplace, uint64_t tb_index, uint64_t tb_len, Edge *dst, uint64_t dst_bias,
                            uint32_t *dst_cnt, uint64_t dst_nleaves) {
  Impl &p = *p_;
  const uint64_t nleaves = src_len >> src_sh;
  // tile size: as large as the workgroup when that still gives every CU several tiles, smaller for smaller windows
  uint32_t tile = p.rb_tile;
  if (tile == 0) {
    tile = kRbTile;
    while (tile > 32 && nleaves / tile < p.rb_min_tiles) tile >>= 1;
  }
  const uint32_t lpc = 64u >> src_sh;
  if (tile < lpc) tile = lpc;
  if (tile > kRbTile) tile = kRbTile;
  const uint64_t ntiles = (nleaves + tile - 1) / tile;
  int rc = ensure_scratch(nleaves);  // d_rank parks the source counts of an in-place window
  if (rc == PPCSR_OK) rc = ensure_tiles(ntiles);
  if (rc != PPCSR_OK) return rc;
  // (inplace: a window of the live array, src_cnt = its slice of the live leaf counts -> the matching slice of the dirty tags)
  const int dsh = nv.g.sh;
  GPU_LAUNCH(p.stream, k_rb_tilesums, ntiles, 256, src_cnt, nleaves, tile, p.d_tiles, inplace ? p.d_rank : (uint32_t *)nullptr,
             inplace ? (uint32_t *)nullptr : dst_cnt, inplace ? (uint64_t)0 : dst_nleaves,
             inplace ? p.v.ldirty + (src_cnt - p.v.leafcnt) : (uint32_t *)nullptr, p.serial);
  // a window that starts at slot 0 crosses a binade per level of the tree: its position table is a serial chain of ~10 us, built
  // INSIDE the scatter launch behind the first tiles when the window is big enough for that to pay (pma_rebalance.h)
  const uint32_t defer = (p.rb_defer_table && tb_index == 0 && tb_len >= (uint64_t)p.rb_defer_table) ? 1u : 0u;
  GPU_LAUNCH(p.stream, k_scan_tilesums, 1, kTileSumThreads, p.d_tiles, ntiles, p.d_total, p.d_table, tb_index, tb_len, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u, defer);
  GPU_LAUNCH(p.stream, k_rb_scatter, ntiles, 256, nv, src_items, src_lo, src_len, src_sh,
               inplace ? (const uint32_t *)p.d_rank : (const uint32_t *)src_cnt, tile, p.rb_prefetch ? 4u : 1u, (const uint32_t *)p.d_tiles,
               p.d_table, dst, dst_bias, dst_cnt, nv.g.sh, (uint64_t)0, defer);
  return PPCSR_OK;
}

// double_list / half_list (PCSR.cpp:251-320): out-of-place whole-array rebalance into a fresh buffer
int Engine::resize(uint64_t newN) {
  Impl &p = *p_;
  if (newN < 2) return fail(PPCSR_EUNSUPPORTED, "array cannot shrink further");
  if (newN > (1ull << 31)) return fail(PPCSR_EUNSUPPORTED, "edge array would exceed 2^31 slots (reference indexes with int)");
  const View old = p.v;
  const uint64_t oldN = old.g.N;
  const uint64_t old_leaves = oldN >> old.g.sh;
  Geometry g;
  compute_geometry(newN, old.g.n, old.g.lock_search, &g);
  g.narrow = old.g.narrow;
  View nv = old;
  nv.g = g;
  const uint64_t new_leaves = newN >> g.sh;
  nv.items = nullptr;
  nv.leafcnt = nullptr;
  DevGuard guard;  // a failure below leaves the engine on its old (intact) arrays
  guard.add(&nv.items);
  guard.add(&nv.leafcnt);
  GCHK(gpu::dmalloc((void **)&nv.items, newN * sizeof(Edge)));
  GCHK(gpu::dmalloc((void **)&nv.leafcnt, new_leaves * sizeof(uint32_t)));
  int rc = PPCSR_OK;
  if (p.time_resize) p.timer.start(p.stream);  // (resize_bench: device time of the passes alone, allocations outside)
  if (p.scatter_variant == 2 && old.g.logN <= 32) {
    rc = rebalance_fused(nv, old.items, 0, oldN, old.g.sh, old.leafcnt, false, 0, newN, nv.items, 0, nv.leafcnt, new_leaves);
    if (rc != PPCSR_OK) return rc;
  } else {
    rc = rank_scan(old.leafcnt, old_leaves, true, 0, newN);
    if (rc != PPCSR_OK) return rc;
    GCHK(gpu::dset(nv.leafcnt, 0, new_leaves * sizeof(uint32_t), p.stream));
  }
  // one fused pass: read the old array once, write every slot of the new array exactly once (elements + nulls)
  if (p.scatter_variant == 2 && old.g.logN <= 32) {
  } else if (p.scatter_variant == 1)
    GPU_LAUNCH(p.stream, k_scatter_runs, grid_for((oldN + 63) / 64, 4, p.scatter_blocks), 256, nv, (const Edge *)old.items, (uint64_t)0, oldN,
               old.g.sh, (const uint32_t *)p.d_rank, (const ChainTable *)p.d_table, nv.items, (uint64_t)0, nv.leafcnt, nv.g.sh,
               (uint64_t)0);
  else
    GPU_LAUNCH(p.stream, k_scatter_fill, grid_for((oldN + 63) / 64, 4), 256, nv, (const Edge *)old.items, (uint64_t)0, oldN,
               old.g.sh, (const uint32_t *)p.d_rank, (const ChainTable *)p.d_table, nv.items, (uint64_t)0, nv.leafcnt, nv.g.sh,
               (uint64_t)0);
  if (p.time_resize) p.timer.stop(p.stream);
  GCHK(gpu::sync(p.stream));
  GCHK(gpu::last_error());
  if (p.time_resize) p.last_resize_ms = p.timer.ms();
  guard.dismiss();
  GPU_DFREE(old.items);
  GPU_DFREE(old.leafcnt);
  {
    View tmp = old;
    free_aux(p, tmp);
  }
  p.v = nv;
  p.v.wres = p.v.rres = p.v.dres = nullptr;
  GCHK(alloc_aux(p, p.v));  // fresh reservation / stamp arrays for the new leaf count
  p.array_gen++;  // (snapshots of the old array are copied in full the next time they are used)
  if (newN > oldN) p.st.double_calls++; else p.st.half_calls++;
  p.st.redistribute_calls++;
  p.st.redistribute_slots += newN;
  return PPCSR_OK;  // (a structure in the sequential regime is re-checked by add_node / apply_batch, not here: the caller may be in the middle of an update)
}

// an in-place rebalance whose tile order was not a valid schedule gave up waiting and said so: the array is not to be trusted
int Engine::inplace_fault_check() {
  Impl &p = *p_;
  if (!p.ip_used) return PPCSR_OK;
  uint32_t err = 0;
  GCHK(gpu::d2h(&err, p.d_ip + 1, sizeof(uint32_t), p.stream));
  GCHK(gpu::sync(p.stream));
  p.ip_used = false;
  if (err) return fail(PPCSR_EINTERNAL, "in-place window rebalance: a tile waited for a source tile that never reported (tile order invalid)");
  return PPCSR_OK;
}

// window rebalance too large for one wave: leaf-rank scan + exact position table + ONE fused scatter/fill pass into a
// persistent scratch array; a whole-array window then just swaps the buffers, a partial window is copied back
int Engine::big_redistribute(uint64_t wstart, uint64_t wlen, bool sync) {
  Impl &p = *p_;
  const View v = p.v;
  const uint64_t leaf_lo = wstart >> v.g.sh, nleaves = wlen >> v.g.sh;
  const bool fused = p.scatter_variant == 2 && v.g.logN <= 32;
  int rc = fused ? ensure_scratch(nleaves) : rank_scan(v.leafcnt + leaf_lo, nleaves, true, wstart, wlen);
  if (rc != PPCSR_OK) return rc;
  const bool whole = (wstart == 0 && wlen == v.g.N);
  if (fused && !whole && p.rb_inplace_min && wlen >= p.rb_inplace_min) {
    // in place: tiles of 2048 (4096) slots held in registers, ordered so that nobody overwrites what has not been read
    const uint32_t cpw = p.rb_inplace_cpw ? p.rb_inplace_cpw : ((wlen / 2048u <= kIpMaxTiles) ? 8u : 16u);
    const uint32_t tile_slots = 4u * cpw * 64u;
    const uint32_t tile_leaves = tile_slots >> v.g.sh;
    const uint64_t ntiles = wlen / tile_slots;
    if (ntiles >= 1 && ntiles <= kIpMaxTiles && wlen % tile_slots == 0 && tile_leaves >= 1 && tile_leaves <= kRbTile && (64u >> v.g.sh) >= 1u) {
      rc = ensure_tiles(ntiles);
      if (rc != PPCSR_OK) return rc;
      if (++p.ip_epoch == 0) {  // (flags are compared with the epoch: clear them when it wraps)
        GCHK(gpu::dset(p.d_ip + kIpHdrWords + kIpMaxTiles, 0, (uint64_t)kIpMaxTiles * sizeof(uint32_t), p.stream));
        p.ip_epoch = 1;
      }
      uint32_t *order = p.d_ip + kIpHdrWords, *flags = p.d_ip + kIpHdrWords + kIpMaxTiles;
      GPU_LAUNCH(p.stream, k_rb_tilesums, ntiles, 256, v.leafcnt + leaf_lo, nleaves, tile_leaves, p.d_tiles, p.d_rank, (uint32_t *)nullptr, (uint64_t)0,
                 v.ldirty + leaf_lo, p.serial);
      GPU_LAUNCH(p.stream, k_scan_tilesums, 1, kTileSumThreads, p.d_tiles, ntiles, p.d_total, p.d_table, wstart, wlen, order, p.d_ip, tile_slots, 0u);
      if (cpw == 8)
        GPU_LAUNCH(p.stream, k_rb_inplace8, ntiles, 256, v, wstart, wlen, v.g.sh, (const uint32_t *)p.d_rank, (const uint32_t *)p.d_tiles,
                   (const ChainTable *)p.d_table, (const uint32_t *)order, p.d_ip, flags, p.ip_epoch, p.ip_lists);
      else
        GPU_LAUNCH(p.stream, k_rb_inplace16, ntiles, 256, v, wstart, wlen, v.g.sh, (const uint32_t *)p.d_rank, (const uint32_t *)p.d_tiles,
                   (const ChainTable *)p.d_table, (const uint32_t *)order, p.d_ip, flags, p.ip_epoch, p.ip_lists);
      p.ip_used = true;
      if (sync) {
        GCHK(gpu::sync(p.stream));
        GCHK(gpu::last_error());
        rc = inplace_fault_check();
        if (rc != PPCSR_OK) return rc;
      }
      p.st.big_redistributes++;
      return PPCSR_OK;
    }
  }
  const uint64_t need = whole ? v.g.N : wlen;
  if (whole ? (p.scratch_cap != need) : (p.scratch_cap < need)) {  // a swapped-in buffer must be exactly N slots
    if (p.d_scratch) GPU_DFREE(p.d_scratch);
    p.d_scratch = nullptr;
    p.scratch_cap = 0;
    GCHK(gpu::dmalloc((void **)&p.d_scratch, need * sizeof(Edge)));
    p.scratch_cap = need;
  }
  if (fused) {
    rc = rebalance_fused(v, v.items, wstart, wlen, v.g.sh, v.leafcnt + leaf_lo, true, wstart, wlen, p.d_scratch, wstart, v.leafcnt, 0);
    if (rc != PPCSR_OK) return rc;
  } else {
    GCHK(gpu::dset(v.leafcnt + leaf_lo, 0, nleaves * sizeof(uint32_t), p.stream));
    GPU_LAUNCH(p.stream, k_fill_u32, grid_for(nleaves, 256), 256, v.ldirty + leaf_lo, nleaves, p.serial);  // dirty tags of the window
  }
  if (fused) {
  } else if (p.scatter_variant == 1)
    GPU_LAUNCH(p.stream, k_scatter_runs, grid_for((wlen + 63) / 64, 4, p.scatter_blocks), 256, v, (const Edge *)v.items, wstart, wlen, v.g.sh,
               (const uint32_t *)p.d_rank, (const ChainTable *)p.d_table, p.d_scratch, wstart, v.leafcnt, v.g.sh, (uint64_t)0);
  else
    GPU_LAUNCH(p.stream, k_scatter_fill, grid_for((wlen + 63) / 64, 4), 256, v, (const Edge *)v.items, wstart, wlen, v.g.sh,
               (const uint32_t *)p.d_rank, (const ChainTable *)p.d_table, p.d_scratch, wstart, v.leafcnt, v.g.sh, (uint64_t)0);
  if (whole) {
    std::swap(p.v.items, p.d_scratch);  // scratch_cap == N: the old array becomes the scratch
  } else {
    GPU_LAUNCH(p.stream, k_copy_slots, grid_for(wlen * 3, 256 * 8), 256, (const Edge *)p.d_scratch, v.items + wstart, wlen);
  }
  if (sync) {
    GCHK(gpu::sync(p.stream));
    GCHK(gpu::last_error());
  }
  p.st.big_redistributes++;
  return PPCSR_OK;
}

// ---- single operations (reference API surface, PCSR.h:73-124) ------------------------------------------------
int Engine::add_edge(uint32_t s, uint32_t d, uint32_t value) {
  if (value == 0) return PPCSR_OK;  // reference: silently ignored (PCSR.cpp:1375)
  Op op{s, d, value};
  return apply_batch_host(&op, 1);
}
int Engine::remove_edge(uint32_t s, uint32_t d) {
  if (s >= n()) return fail(PPCSR_EINVAL, "remove_edge: src out of range (undefined behaviour in the reference)");
  Op op{s, d, 0};
  return apply_batch_host(&op, 1);
}

int Engine::add_node() {  // PCSR.cpp:681-703
  Impl &p = *p_;
  GCHK(gpu::set_device(device_));
  const uint32_t len = p.v.g.n;
  if ((uint64_t)len + 1 > p.n_cap) {
    const uint64_t ncap = p.n_cap * 2;
    Node *nn = nullptr;
    GCHK(gpu::dmalloc((void **)&nn, ncap * sizeof(Node)));
    GCHK(gpu::d2d(nn, p.v.nodes, (uint64_t)len * sizeof(Node), p.stream));
    GCHK(gpu::sync(p.stream));
    GPU_DFREE(p.v.nodes);
    p.v.nodes = nn;
    p.n_cap = ncap;
    GCHK(alloc_vertex_aux(p, p.v));
  }
  Node nd;
  uint32_t sval = len;
  if (len > 0) {
    Node last;
    GCHK(gpu::d2h(&last, p.v.nodes + (len - 1), sizeof(Node), p.stream));
    GCHK(gpu::sync(p.stream));
    nd.beginning = last.end;
    nd.end = nd.beginning + 1;
  } else {
    nd.beginning = 0;
    nd.end = 1;
    sval = kMax;
  }
  nd.num_neighbors = 0;
  GCHK(gpu::h2d(p.v.nodes + len, &nd, sizeof(Node), p.stream));
  GCHK(gpu::sync(p.stream));
  p.v.g.n = len + 1;
  p.array_gen++;  // (the vertex set changed: snapshots are copied in full the next time)
  Op op{len, nd.beginning, sval};
  const int rc = run_exclusive(op, XF_ADD_NODE | XF_FORCE_NOINFO);
  if (rc != PPCSR_OK) {
    p.v.g.n = len;  // the vertex was not added
    return rc;
  }
  return recheck_ranges();
}

int Engine::edge_exists(uint32_t s, uint32_t d, int *out) {
  Impl &p = *p_;
  if (s >= n()) return fail(PPCSR_EINVAL, "edge_exists: src out of range");
  GCHK(gpu::set_device(device_));
  GPU_LAUNCH(p.stream, k_edge_exists, 1, 64, p.v, s, d, p.d_xout);
  GCHK(gpu::d2h(p.h_xout, p.d_xout, sizeof(ExclOut), p.stream));
  GCHK(gpu::sync(p.stream));
  *out = (int)p.h_xout->found;
  return PPCSR_OK;
}

int Engine::get_node(uint32_t vtx, Node *out) {
  Impl &p = *p_;
  if (vtx >= n()) return fail(PPCSR_EINVAL, "get_node: vertex out of range");
  GCHK(gpu::set_device(device_));
  GCHK(gpu::d2h(out, p.v.nodes + vtx, sizeof(Node), p.stream));
  GCHK(gpu::sync(p.stream));
  return PPCSR_OK;
}

int Engine::get_neighbourhood(int src, int *out, uint64_t cap, uint64_t *count) {
  Impl &p = *p_;
  *count = 0;
  if (src < 0 || (uint64_t)src >= n()) return PPCSR_OK;  // reference returns an empty vector (PCSR.cpp:903)
  GCHK(gpu::set_device(device_));
  if (cap > p.nbr_cap) {
    if (p.d_nbr) GPU_DFREE(p.d_nbr);
    p.d_nbr = nullptr;
    p.nbr_cap = 0;
    GCHK(gpu::dmalloc((void **)&p.d_nbr, cap * sizeof(int)));
    p.nbr_cap = cap;
  }
  GPU_LAUNCH(p.stream, k_neighbourhood, 1, 64, p.v, (uint32_t)src, (out && cap) ? p.d_nbr : (int *)nullptr, cap, p.d_total);
  GCHK(gpu::d2h(p.h_total, p.d_total, sizeof(unsigned long long), p.stream));
  GCHK(gpu::sync(p.stream));
  *count = *p.h_total;
  const uint64_t m = std::min<uint64_t>(*count, cap);
  if (out && m) {
    GCHK(gpu::d2h(out, p.d_nbr, m * sizeof(int), p.stream));
    GCHK(gpu::sync(p.stream));
  }
  return PPCSR_OK;
}

int Engine::read_neighbourhood(int src) {
  uint64_t c = 0;
  return get_neighbourhood(src, nullptr, 0, &c);
}

// bulk neighbour scan: live-edge count per 64-slot chunk from the leaf counts and the sentinel positions (no pass over
// the edge array), exclusive scan, then ONE streaming pass that writes dests (array order == CSR order) and row offsets.
// (A single-kernel decoupled look-back variant was tried and measured slower on MI355X — 110 us vs 85 us at N = 2^24,
//  1.5 K polling workgroups disturb the streaming loads — and was removed; see the git history, "one-pass bulk neighbour scan".)
int Engine::scan_launch(unsigned long long *d_rows, int *d_dst, uint64_t cap, const float *d_values, float *d_contrib, Op *d_triples,
                        uint32_t src_base) {
  Impl &p = *p_;
  const uint64_t N = p.v.g.N, nchunks = (N + 63) / 64;
  bool fresh_state = (p.scan_nchunks != nchunks);  // the sentinel counts are laid out (and left zeroed) per array size
  p.scan_nchunks = nchunks;
  if (p.scan_state_cap < 2 * nchunks) {
    if (p.d_scan_state) GPU_DFREE(p.d_scan_state);
    p.d_scan_state = nullptr;
    p.scan_state_cap = 0;
    GCHK(gpu::dmalloc((void **)&p.d_scan_state, 2 * nchunks * sizeof(uint32_t) + 64));
    p.scan_state_cap = 2 * nchunks;
  }
  uint32_t *d_cs = reinterpret_cast<uint32_t *>(p.d_scan_state), *d_cc = d_cs + nchunks;
  // tile of chunks per workgroup: as large as the workgroup while the array still yields thousands of tiles
  uint32_t tile = 256;
  while (tile > 16 && nchunks / tile < p.rb_min_tiles) tile >>= 1;
  const uint64_t ntiles = (nchunks + tile - 1) / tile;
  int rc = ensure_tiles(ntiles);
  if (rc != PPCSR_OK) return rc;
  if (fresh_state) GCHK(gpu::dset(d_cs, 0, nchunks * sizeof(uint32_t), p.stream));  // later scans find it zeroed (k_chunk_counts)
  GPU_LAUNCH(p.stream, k_chunk_sentinels, grid_for(n(), 256), 256, p.v, d_cs);
  GPU_LAUNCH(p.stream, k_chunk_counts, ntiles, 256, p.v, d_cs, d_cc, tile, p.d_tiles);
  GPU_LAUNCH(p.stream, k_scan_tilesums, 1, kTileSumThreads, p.d_tiles, ntiles, p.d_total, (ChainTable *)nullptr, (uint64_t)0, (uint64_t)0, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u, 0u);
  GPU_LAUNCH(p.stream, k_scan_write, ntiles, 256, p.v, (const uint32_t *)d_cc, tile, (const uint32_t *)p.d_tiles, d_rows, d_dst, cap,
             d_values, d_contrib, d_triples, src_base);
  return PPCSR_OK;
}

// every edge as (src + src_base, dest, value) in array order (= ascending src, then ascending dest) into d_out (device memory,
// room for `cap` records; nullptr / 0 only counts).  *total = edges held.  What pppcsr_repartition ships between partitions.
int Engine::export_num_neighbors_device(uint32_t base, Op *d_out) {
  Impl &p = *p_;
  GCHK(gpu::set_device(device_));
  if (n() == 0) return PPCSR_OK;
  GPU_LAUNCH(p.stream, k_nn_export, grid_for(n(), 256), 256, p.v, base, d_out);
  GCHK(gpu::sync(p.stream));
  GCHK(gpu::last_error());
  return PPCSR_OK;
}
int Engine::set_num_neighbors_device(const Op *d_recs, uint64_t cnt) {
  Impl &p = *p_;
  GCHK(gpu::set_device(device_));
  if (cnt == 0) return PPCSR_OK;
  GPU_LAUNCH(p.stream, k_nn_set, grid_for(cnt, 256), 256, p.v, d_recs, cnt);
  GCHK(gpu::sync(p.stream));
  GCHK(gpu::last_error());
  return PPCSR_OK;
}
int Engine::export_triples_device(uint32_t src_base, Op *d_out, uint64_t cap, uint64_t *total) {
  Impl &p = *p_;
  GCHK(gpu::set_device(device_));
  int rc = scan_launch(nullptr, nullptr, d_out ? cap : 0, nullptr, nullptr, d_out, src_base);
  if (rc != PPCSR_OK) return rc;
  GCHK(gpu::d2h(p.h_total, p.d_total, sizeof(unsigned long long), p.stream));
  // the last slot belongs to no neighbourhood (the last vertex's `end` is N - 1, PCSR.cpp:87) and the scan leaves it out,
  // but it can hold an edge (a slide can push one there; the next insert at it doubles the array, PCSR.cpp:992-997)
  Edge last = null_edge();
  GCHK(gpu::d2h(&last, p.v.items + (p.v.g.N - 1), sizeof(Edge), p.stream));
  GCHK(gpu::sync(p.stream));
  GCHK(gpu::last_error());
  uint64_t tot = *p.h_total;
  if (last.value != 0 && !is_sentinel(last)) {
    if (d_out && tot < cap) {
      const Op o{last.src + src_base, last.dest, last.value};
      GCHK(gpu::h2d(d_out + tot, &o, sizeof(Op), p.stream));
      GCHK(gpu::sync(p.stream));
    }
    tot++;
  }
  if (total) *total = tot;
  return (d_out && tot > cap) ? PPCSR_ERANGE : PPCSR_OK;
}

// ---- bulk build (kernels k_bb_*) -----------------------------------------------------------------------
static int sort_edges_stable(gpu::stream_t st, unsigned long long *kin, unsigned long long *kout, uint32_t *vin, uint32_t *vout,
                             uint64_t m, unsigned bits) {
#if defined(PPCSR_SIM)
  (void)st;
  (void)bits;
  std::vector<uint64_t> idx(m);
  for (uint64_t i = 0; i < m; i++) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](uint64_t a, uint64_t b) { return kin[a] < kin[b]; });
  for (uint64_t i = 0; i < m; i++) {
    kout[i] = kin[idx[i]];
    vout[i] = vin[idx[i]];
  }
  return 0;
#else
  size_t tmp_bytes = 0;
  if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, kin, kout, vin, vout, (size_t)m, 0u, bits, st) != hipSuccess) return 3;
  void *tmp = nullptr;
  if (hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1) != hipSuccess) return 2;
  const hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, (size_t)m, 0u, bits, st);
  (void)hipStreamSynchronize(st);
  (void)hipFree(tmp);
  return e == hipSuccess ? 0 : 3;
#endif
}

int Engine::bulk_build(const Op *host_ops, uint64_t m, double *device_ms) { return bulk_build_from(host_ops, false, m, device_ms); }
int Engine::bulk_build_device(const Op *d_adds, uint64_t m, double *device_ms) { return bulk_build_from(d_adds, true, m, device_ms); }
int Engine::bulk_build_from(const Op *in_ops, bool on_device, uint64_t m, double *device_ms) {
  Impl &p = *p_;
  GCHK(gpu::set_device(device_));
  const uint32_t nn = n();
  if (nn == 0) return fail(PPCSR_EINVAL, "bulk_build: the graph has no vertices");
  if (m >= (1ull << 31)) return fail(PPCSR_EUNSUPPORTED, "bulk_build: too many edges for one call");
  {  // only an EMPTY graph can be bulk-built (everything live must be a sentinel)
    int rc = rank_scan(p.v.leafcnt, p.v.g.N >> p.v.g.sh, false, 0, 0);
    if (rc != PPCSR_OK) return rc;
    GCHK(gpu::d2h(p.h_total, p.d_total, sizeof(unsigned long long), p.stream));
    GCHK(gpu::sync(p.stream));
    if (*p.h_total != (unsigned long long)nn) return fail(PPCSR_EINVAL, "bulk_build: the graph already holds edges");
  }
  Op *d_ops = nullptr;
  unsigned long long *d_k0 = nullptr, *d_k1 = nullptr;
  uint32_t *d_v0 = nullptr, *d_v1 = nullptr, *d_flags = nullptr;
  DevGuard tmpg;
  tmpg.add(&d_ops); tmpg.add(&d_k0); tmpg.add(&d_k1); tmpg.add(&d_v0); tmpg.add(&d_v1); tmpg.add(&d_flags);
  const uint64_t mm = std::max<uint64_t>(m, 1);
  if (!on_device) GCHK(gpu::dmalloc((void **)&d_ops, mm * sizeof(Op)));
  GCHK(gpu::dmalloc((void **)&d_k0, mm * sizeof(unsigned long long)));
  GCHK(gpu::dmalloc((void **)&d_k1, mm * sizeof(unsigned long long)));
  GCHK(gpu::dmalloc((void **)&d_v0, mm * sizeof(uint32_t)));
  GCHK(gpu::dmalloc((void **)&d_v1, mm * sizeof(uint32_t)));
  GCHK(gpu::dmalloc((void **)&d_flags, mm * sizeof(uint32_t)));
  if (m && !on_device) GCHK(gpu::h2d(d_ops, in_ops, m * sizeof(Op), p.stream));
  const Op *src_ops = on_device ? in_ops : (const Op *)d_ops;
  p.timer.start(p.stream);
  uint64_t E = 0;
  if (m) {
    GPU_LAUNCH(p.stream, k_bb_keys, grid_for(m, 256), 256, src_ops, m, nn, d_k0, d_v0);
    unsigned bits = 1;  // the source half of the key never exceeds n (entries to ignore carry src = n)
    while (bits < 32 && ((uint64_t)nn >> bits) != 0) bits++;
    int rc = sort_edges_stable(p.stream, d_k0, d_k1, d_v0, d_v1, m, 32u + bits);
    if (rc != 0) return fail(rc == 2 ? PPCSR_ENOMEM : PPCSR_EHIP, "bulk_build: device sort failed");
    GPU_LAUNCH(p.stream, k_bb_flags, grid_for(m, 256), 256, (const unsigned long long *)d_k1, m, nn, d_flags);
    rc = rank_scan(d_flags, m, false, 0, 0);
    if (rc != PPCSR_OK) return rc;
    GCHK(gpu::d2h(p.h_total, p.d_total, sizeof(unsigned long long), p.stream));
    GCHK(gpu::sync(p.stream));
    GCHK(gpu::last_error());
    E = *p.h_total;
  } else {
    const unsigned long long zero = 0;
    GCHK(gpu::h2d(p.d_total, &zero, sizeof(zero), p.stream));
  }
  // array size: the smallest power of two (not below the current one) whose root still accepts one more insert
  const uint64_t j = (uint64_t)nn + E;
  const View old = p.v;
  uint64_t newN = old.g.N;
  Geometry g;
  for (;;) {
    compute_geometry(newN, nn, old.g.lock_search, &g);
    g.narrow = old.g.narrow;
    if (j + 1 < (uint64_t)g.t_up[0] || newN >= (1ull << 31)) break;
    newN *= 2;
  }
  if (j + 1 >= (uint64_t)g.t_up[0]) return fail(PPCSR_EUNSUPPORTED, "bulk_build: edge array would exceed 2^31 slots");
  View nv = old;
  nv.g = g;
  nv.items = nullptr;
  nv.leafcnt = nullptr;
  DevGuard nvg;
  nvg.add(&nv.items);
  nvg.add(&nv.leafcnt);
  GCHK(gpu::dmalloc((void **)&nv.items, newN * sizeof(Edge)));
  GCHK(gpu::dmalloc((void **)&nv.leafcnt, (newN >> g.sh) * sizeof(uint32_t)));
  ChainTable *htb = new ChainTable;
  build_chain_table(0, newN, j, htb);
  const bool overflow = htb->overflow != 0;
  GCHK(gpu::h2d(p.d_table, htb, sizeof(ChainTable), p.stream));
  GCHK(gpu::sync(p.stream));
  delete htb;
  if (overflow) return fail(PPCSR_EINTERNAL, "bulk_build: position table overflow");
  GPU_LAUNCH(p.stream, k_fill_null, grid_for(newN * 3, 256 * 8), 256, nv.items, (uint64_t)0, newN);
  GPU_LAUNCH(p.stream, k_bb_vertices, grid_for(nn, 256), 256, nv, (const unsigned long long *)d_k1, m, (const uint32_t *)p.d_rank,
             (const unsigned long long *)p.d_total, (const ChainTable *)p.d_table);
  if (m)
    GPU_LAUNCH(p.stream, k_bb_edges, grid_for(m, 256), 256, nv, (const unsigned long long *)d_k1, (const uint32_t *)d_v1,
               (const uint32_t *)d_flags, (const uint32_t *)p.d_rank, m, (const ChainTable *)p.d_table);
  GPU_LAUNCH(p.stream, k_recount, grid_for((newN + 63) / 64, 4), 256, nv, (uint64_t)0, newN);
  p.timer.stop(p.stream);
  GCHK(gpu::sync(p.stream));
  GCHK(gpu::last_error());
  if (device_ms) *device_ms = p.timer.ms();
  nvg.dismiss();
  GPU_DFREE(old.items);
  GPU_DFREE(old.leafcnt);
  {
    View tmp = old;
    free_aux(p, tmp);
  }
  p.v = nv;
  p.v.wres = p.v.rres = p.v.dres = nullptr;
  GCHK(alloc_aux(p, p.v));
  p.array_gen++;
  return PPCSR_OK;
}

// ---- consumers ------------------------------------------------------------------------------------
// bfs.h:15-36: level of every vertex from `start` (UINT32_MAX = unreachable); one launch per level over the gapped array
int Engine::bfs(uint32_t start, uint32_t *levels, double *device_ms) {
  Impl &p = *p_;
  GCHK(gpu::set_device(device_));
  const uint32_t nn = n();
  if (start >= nn) return fail(PPCSR_EINVAL, "bfs: start vertex out of range");
  uint32_t *d_lv = nullptr, *d_f0 = nullptr, *d_f1 = nullptr, *d_cnt = nullptr, *d_fb = nullptr, *d_vb = nullptr;
  DevGuard tmpg;
  tmpg.add(&d_lv); tmpg.add(&d_f0); tmpg.add(&d_f1); tmpg.add(&d_cnt); tmpg.add(&d_fb); tmpg.add(&d_vb);
  const uint64_t bit_words = ((uint64_t)nn + 63) / 64 * 2;  // frontier / visited bitmaps of the streaming levels
  GCHK(gpu::dmalloc((void **)&d_fb, bit_words * sizeof(uint32_t)));
  GCHK(gpu::dmalloc((void **)&d_vb, bit_words * sizeof(uint32_t)));
  GCHK(gpu::dmalloc((void **)&d_lv, (uint64_t)nn * sizeof(uint32_t)));
  GCHK(gpu::dmalloc((void **)&d_f0, (uint64_t)nn * sizeof(uint32_t)));
  GCHK(gpu::dmalloc((void **)&d_f1, (uint64_t)nn * sizeof(uint32_t)));
  // [0] vertices found, [1] a hub was left to the streaming pass, [kBfsStripeWords...] the streaming pass's striped count
  constexpr uint32_t cnt_words = (kBfsStripes + 1) * kBfsStripeWords;
  GCHK(gpu::dmalloc((void **)&d_cnt, cnt_words * sizeof(uint32_t)));
  p.timer.start(p.stream);
  GCHK(gpu::dset(d_lv, 0xFF, (uint64_t)nn * sizeof(uint32_t), p.stream));
  const uint32_t zero = 0;
  GCHK(gpu::h2d(d_lv + start, &zero, sizeof(uint32_t), p.stream));
  GCHK(gpu::h2d(d_f0, &start, sizeof(uint32_t), p.stream));
  // Hybrid: a small frontier is expanded one wave per vertex (k_bfs_level, builds the next frontier list); a frontier that
  // is a sizeable share of the graph is expanded by one streaming pass over the whole gapped array (k_bfs_edges) — its cost
  // does not depend on hub degrees — and the list is rebuilt only when the frontier becomes small again.
  uint32_t nfront = 1, level = 0;
  uint32_t *cur = d_f0, *nxt = d_f1;
  std::vector<uint32_t> h_cnt(cnt_words, 0);
  auto striped = [&]() {
    uint32_t sum = 0;
    for (uint32_t k = 1; k <= kBfsStripes; k++) sum += h_cnt[k * kBfsStripeWords];
    return sum;
  };
  bool have_list = true;
  const uint64_t N = p.v.g.N;
  const uint32_t big = (uint32_t)std::max<uint64_t>(64, (uint64_t)nn / 256);  // frontier size from which the pass is cheaper
  while (nfront > 0) {
    GCHK(gpu::dset(d_cnt, 0, cnt_words * sizeof(uint32_t), p.stream));
    if (nfront >= big) {
      GPU_LAUNCH(p.stream, k_bfs_bits, grid_for(nn, 256, 4096), 256, (const uint32_t *)d_lv, nn, level, d_fb, d_vb);
      GPU_LAUNCH(p.stream, k_bfs_edges_bits, grid_for((N + 255) / 256, 4, 8192), 256, p.v, level, (const uint32_t *)d_fb, (const uint32_t *)d_vb, d_lv, d_cnt + kBfsStripeWords);
      have_list = false;
    } else {
      if (!have_list) {  // (the pass only counted claims — an upper bound; the list gives the exact frontier)
        GPU_LAUNCH(p.stream, k_bfs_collect, grid_for(nn, 256), 256, (const uint32_t *)d_lv, nn, level, cur, d_cnt);
        GCHK(gpu::d2h(h_cnt.data(), d_cnt, sizeof(uint32_t), p.stream));
        GCHK(gpu::sync(p.stream));
        nfront = h_cnt[0];
        GCHK(gpu::dset(d_cnt, 0, 2 * sizeof(uint32_t), p.stream));
      }
      GPU_LAUNCH(p.stream, k_bfs_level, grid_for(nfront, 4, 16384), 256, p.v, (const uint32_t *)cur, nfront, level, d_lv, nxt, d_cnt);
      have_list = true;
      std::swap(cur, nxt);
    }
    GCHK(gpu::d2h(h_cnt.data(), d_cnt, cnt_words * sizeof(uint32_t), p.stream));
    GCHK(gpu::sync(p.stream));
    GCHK(gpu::last_error());
    if (h_cnt[1]) {  // hubs of this level were skipped by the per-vertex kernel: one pass finishes the level
      GPU_LAUNCH(p.stream, k_bfs_bits, grid_for(nn, 256, 4096), 256, (const uint32_t *)d_lv, nn, level, d_fb, d_vb);
      GPU_LAUNCH(p.stream, k_bfs_edges_bits, grid_for((N + 255) / 256, 4, 8192), 256, p.v, level, (const uint32_t *)d_fb, (const uint32_t *)d_vb, d_lv, d_cnt + kBfsStripeWords);
      GCHK(gpu::d2h(h_cnt.data(), d_cnt, cnt_words * sizeof(uint32_t), p.stream));
      GCHK(gpu::sync(p.stream));
      GCHK(gpu::last_error());
      have_list = false;
    }
    nfront = h_cnt[0] + striped();
    level++;
  }
  p.timer.stop(p.stream);
  GCHK(gpu::d2h(levels, d_lv, (uint64_t)nn * sizeof(uint32_t), p.stream));
  GCHK(gpu::sync(p.stream));
  if (device_ms) *device_ms = p.timer.ms();
  return PPCSR_OK;
}

// stable sort of (key, value) pairs by key: rocPRIM's radix sort on the device, std::stable_sort in the CPU emulator
static int sort_pairs_stable(gpu::stream_t st, uint32_t *kin, uint32_t *kout, float *vin, float *vout, uint64_t m, unsigned bits) {
#if defined(PPCSR_SIM)
  (void)st;
  (void)bits;
  std::vector<uint64_t> idx(m);
  for (uint64_t i = 0; i < m; i++) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](uint64_t a, uint64_t b) { return kin[a] < kin[b]; });
  for (uint64_t i = 0; i < m; i++) {
    kout[i] = kin[idx[i]];
    vout[i] = vin[idx[i]];
  }
  return 0;
#else
  size_t tmp_bytes = 0;
  if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, kin, kout, vin, vout, (size_t)m, 0u, bits, st) != hipSuccess) return 3;
  void *tmp = nullptr;
  if (hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1) != hipSuccess) return 2;
  const hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, (size_t)m, 0u, bits, st);
  (void)hipStreamSynchronize(st);
  (void)hipFree(tmp);
  return e == hipSuccess ? 0 : 3;
#endif
}

// pagerank.h:15-29: out[d] = sum over edges (s, d), in ascending s, of node_values[s] / num_neighbors(s).  The bulk scan
// emits (dest, contribution) per edge in CSR order, a STABLE sort by dest keeps ascending source order inside every
// destination, and one thread per destination adds its run sequentially: the reference's order of fp32 additions.
int Engine::pagerank(const float *node_values, float *out, double *device_ms) {
  Impl &p = *p_;
  GCHK(gpu::set_device(device_));
  const uint32_t nn = n();
  const uint64_t N = p.v.g.N;
  float *d_val = nullptr, *d_c0 = nullptr, *d_c1 = nullptr, *d_out = nullptr;
  uint32_t *d_k0 = nullptr, *d_k1 = nullptr;
  uint32_t *d_long = nullptr;  // [0]: count, [1..]: destinations with long runs
  DevGuard tmpg;
  tmpg.add(&d_val); tmpg.add(&d_c0); tmpg.add(&d_c1); tmpg.add(&d_out); tmpg.add(&d_k0); tmpg.add(&d_k1); tmpg.add(&d_long);
  GCHK(gpu::dmalloc((void **)&d_val, (uint64_t)nn * sizeof(float)));
  GCHK(gpu::dmalloc((void **)&d_out, (uint64_t)nn * sizeof(float)));
  GCHK(gpu::dmalloc((void **)&d_k0, N * sizeof(uint32_t)));
  GCHK(gpu::dmalloc((void **)&d_k1, N * sizeof(uint32_t)));
  GCHK(gpu::dmalloc((void **)&d_c0, N * sizeof(float)));
  GCHK(gpu::dmalloc((void **)&d_c1, N * sizeof(float)));
  GCHK(gpu::h2d(d_val, node_values, (uint64_t)nn * sizeof(float), p.stream));
  p.timer.start(p.stream);
  int rc = scan_launch(nullptr, reinterpret_cast<int *>(d_k0), N, d_val, d_c0);
  if (rc != PPCSR_OK) return rc;
  GCHK(gpu::d2h(p.h_total, p.d_total, sizeof(unsigned long long), p.stream));
  GCHK(gpu::sync(p.stream));
  GCHK(gpu::last_error());
  const uint64_t m = *p.h_total;
  if (m) {
    unsigned bits = 1;  // keys are clamped to [0, n]
    while (bits < 32 && ((uint64_t)nn >> bits) != 0) bits++;
    rc = sort_pairs_stable(p.stream, d_k0, d_k1, d_c0, d_c1, m, bits);
    if (rc != 0) return fail(rc == 2 ? PPCSR_ENOMEM : PPCSR_EHIP, "pagerank: device sort failed");
  }
  GCHK(gpu::dmalloc((void **)&d_long, ((uint64_t)nn + 1) * sizeof(uint32_t)));
  GCHK(gpu::dset(d_long, 0, sizeof(uint32_t), p.stream));
  GPU_LAUNCH(p.stream, k_pr_segsum, grid_for(nn, 256), 256, (const uint32_t *)d_k1, (const float *)d_c1, m, nn, d_out, d_long + 1, d_long);
  GPU_LAUNCH(p.stream, k_pr_longruns, 2048, 256, (const uint32_t *)d_k1, (const float *)d_c1, m, (const uint32_t *)(d_long + 1),
             (const uint32_t *)d_long, d_out);
  p.timer.stop(p.stream);
  GCHK(gpu::d2h(out, d_out, (uint64_t)nn * sizeof(float), p.stream));
  GCHK(gpu::sync(p.stream));
  GCHK(gpu::last_error());
  if (device_ms) *device_ms = p.timer.ms();
  return PPCSR_OK;
}

int Engine::scan_all_device(double *ms, uint64_t *total) {
  Impl &p = *p_;
  GCHK(gpu::set_device(device_));
  const uint64_t N = p.v.g.N;
  unsigned long long *d_rows = nullptr;
  int *d_dst = nullptr;
  DevGuard tmpg;
  tmpg.add(&d_rows); tmpg.add(&d_dst);
  GCHK(gpu::dmalloc((void **)&d_rows, ((uint64_t)n() + 1) * sizeof(unsigned long long)));
  GCHK(gpu::dmalloc((void **)&d_dst, N * sizeof(int)));
  int rc = scan_launch(d_rows, d_dst, N);  // warm-up (sizes the tile-state array)
  if (rc != PPCSR_OK) return rc;
  p.timer.start(p.stream);
  rc = scan_launch(d_rows, d_dst, N);
  if (rc != PPCSR_OK) return rc;
  p.timer.stop(p.stream);
  GCHK(gpu::d2h(p.h_total, p.d_total, sizeof(unsigned long long), p.stream));
  GCHK(gpu::sync(p.stream));
  GCHK(gpu::last_error());
  if (ms) *ms = p.timer.ms();
  if (total) *total = *p.h_total;
  return PPCSR_OK;
}

int Engine::scan_all(uint64_t *row_offsets, int *dests, uint64_t cap, uint64_t *total) {
  Impl &p = *p_;
  GCHK(gpu::set_device(device_));
  const uint32_t nn = n();
  unsigned long long *d_rows = nullptr;
  int *d_dst = nullptr;
  DevGuard tmpg;
  tmpg.add(&d_rows); tmpg.add(&d_dst);
  GCHK(gpu::dmalloc((void **)&d_rows, ((uint64_t)nn + 1) * sizeof(unsigned long long)));
  GCHK(gpu::dmalloc((void **)&d_dst, std::max<uint64_t>(cap, 1) * sizeof(int)));
  int rc = scan_launch(d_rows, d_dst, cap);
  if (rc != PPCSR_OK) return rc;
  GCHK(gpu::d2h(p.h_total, p.d_total, sizeof(unsigned long long), p.stream));
  GCHK(gpu::sync(p.stream));
  GCHK(gpu::last_error());
  const uint64_t tot = *p.h_total;
  if (total) *total = tot;
  if (row_offsets && nn) {
    GCHK(gpu::d2h(row_offsets, d_rows, (uint64_t)nn * sizeof(unsigned long long), p.stream));
    GCHK(gpu::sync(p.stream));
  }
  if (row_offsets) row_offsets[nn] = tot;
  if (dests && cap) {
    GCHK(gpu::d2h(dests, d_dst, std::min(cap, tot) * sizeof(int), p.stream));
    GCHK(gpu::sync(p.stream));
  }
  return (tot > cap && dests) ? PPCSR_ERANGE : PPCSR_OK;
}

int Engine::export_state(Edge *items, Node *nodes) {
  Impl &p = *p_;
  GCHK(gpu::set_device(device_));
  if (items) GCHK(gpu::d2h(items, p.v.items, p.v.g.N * sizeof(Edge), p.stream));
  if (nodes && p.v.g.n) GCHK(gpu::d2h(nodes, p.v.nodes, (uint64_t)p.v.g.n * sizeof(Node), p.stream));
  GCHK(gpu::sync(p.stream));
  return PPCSR_OK;
}

int Engine::check_invariants(uint64_t *bad) {
  Impl &p = *p_;
  GCHK(gpu::set_device(device_));
  const uint64_t N = p.v.g.N, leaves = N >> p.v.g.sh;
  std::vector<uint32_t> cnt(leaves), cnt2(leaves);
  GCHK(gpu::d2h(cnt.data(), p.v.leafcnt, leaves * sizeof(uint32_t), p.stream));
  GCHK(gpu::sync(p.stream));
  View tmp = p.v;
  GCHK(gpu::dmalloc((void **)&tmp.leafcnt, leaves * sizeof(uint32_t)));
  GPU_LAUNCH(p.stream, k_recount, grid_for((N + 63) / 64, 4), 256, tmp, (uint64_t)0, N);
  GCHK(gpu::d2h(cnt2.data(), tmp.leafcnt, leaves * sizeof(uint32_t), p.stream));
  GCHK(gpu::sync(p.stream));
  GPU_DFREE(tmp.leafcnt);
  uint64_t b = 0;
  for (uint64_t i = 0; i < leaves; i++) b += cnt[i] != cnt2[i];
  *bad = b;
  return PPCSR_OK;
}

int Engine::pull_stats() {
  Impl &p = *p_;
  GCHK(gpu::set_device(device_));
  GCHK(gpu::d2h(p.h_stats, p.d_stats, kStatShards * sizeof(StatShard), p.stream));
  GCHK(gpu::sync(p.stream));
  return PPCSR_OK;
}

int Engine::stats(EngineStats *out) {
  Impl &p = *p_;
  int rc = pull_stats();
  if (rc != PPCSR_OK) return rc;
  EngineStats s = p.st;
  s.N = p.v.g.N;
  s.n = p.v.g.n;
  s.logN = p.v.g.logN;
  s.H = p.v.g.H;
  s.narrow = p.v.g.narrow;
  for (int i = 0; i < kStatShards; i++) {
    const StatShard &h = p.h_stats[i];
    s.redistribute_calls += h.redistribute_calls;
    s.redistribute_slots += h.redistribute_slots;
    s.not_found += h.not_found;
    s.duplicates += h.duplicates;
    s.noops += h.noops;
    s.slide_slots += h.slide_slots;
  }
  *out = s;
  return PPCSR_OK;
}

static int snap_full_save(Engine::Impl &p, Engine::Impl::Snap &sn) {
  const uint64_t N = p.v.g.N, leaves = N >> p.v.g.sh;
  int e;
  if (sn.cap_slots < N) {
    if (sn.v.items) GPU_DFREE(sn.v.items);
    if (sn.v.leafcnt) GPU_DFREE(sn.v.leafcnt);
    sn.v.items = nullptr;
    sn.v.leafcnt = nullptr;
    sn.cap_slots = 0;
    if ((e = gpu::dmalloc((void **)&sn.v.items, N * sizeof(Edge)))) return e;
    if ((e = gpu::dmalloc((void **)&sn.v.leafcnt, N * sizeof(uint32_t) / 2 + 64))) return e;  // leaves <= N/2
    sn.cap_slots = N;
  }
  if (sn.cap_nodes < p.n_cap) {
    if (sn.v.nodes) GPU_DFREE(sn.v.nodes);
    sn.v.nodes = nullptr;
    sn.cap_nodes = 0;
    if ((e = gpu::dmalloc((void **)&sn.v.nodes, p.n_cap * sizeof(Node)))) return e;
    sn.cap_nodes = p.n_cap;
  }
  sn.v.g = p.v.g;
  if ((e = gpu::d2d(sn.v.items, p.v.items, N * sizeof(Edge), p.stream))) return e;
  if (p.v.g.n && (e = gpu::d2d(sn.v.nodes, p.v.nodes, (uint64_t)p.v.g.n * sizeof(Node), p.stream))) return e;
  if ((e = gpu::d2d(sn.v.leafcnt, p.v.leafcnt, leaves * sizeof(uint32_t), p.stream))) return e;
  sn.valid = true;
  return 0;
}
static int snap_full_load(Engine::Impl &p, Engine::Impl::Snap &sn) {
  const uint64_t N = sn.v.g.N, leaves = N >> sn.v.g.sh;
  int e;
  if (p.v.g.N != N) {  // the array was resized since the snapshot: go back to buffers of the old size
    GPU_DFREE(p.v.items);
    GPU_DFREE(p.v.leafcnt);
    free_aux(p, p.v);
    if ((e = gpu::dmalloc((void **)&p.v.items, N * sizeof(Edge)))) return e;
    if ((e = gpu::dmalloc((void **)&p.v.leafcnt, leaves * sizeof(uint32_t)))) return e;
    p.v.g = sn.v.g;
    if ((e = alloc_aux(p, p.v))) return e;
  }
  p.v.g = sn.v.g;
  if ((e = gpu::d2d(p.v.items, sn.v.items, N * sizeof(Edge), p.stream))) return e;
  if (sn.v.g.n && (e = gpu::d2d(p.v.nodes, sn.v.nodes, (uint64_t)sn.v.g.n * sizeof(Node), p.stream))) return e;
  if ((e = gpu::d2d(p.v.leafcnt, sn.v.leafcnt, leaves * sizeof(uint32_t), p.stream))) return e;
  return 0;
}
static void advance_serial(Engine::Impl &p, uint32_t by) {
  p.serial += by;
  p.v.serial = p.serial;
}
static bool snap_in_step(const Engine::Impl &p, const Engine::Impl::Snap &sn) {
  return sn.valid && sn.gen == p.array_gen && sn.v.g.N == p.v.g.N && sn.v.g.n == p.v.g.n && sn.v.g.logN == p.v.g.logN && p.serial < 0xFFFFFF00u;
}
// live state -> snapshot.  In step with the arrays (same generation): only what was written since its last synchronisation
// (dirty tags); otherwise a full copy.
static int snap_commit(Engine::Impl &p, Engine::Impl::Snap &sn) {
  int e;
  if (p.serial >= 0xFFFFFF00u) {  // (4 G synchronisations: start the tags over; every snapshot is copied in full once)
    const uint64_t leaves = p.v.g.N >> p.v.g.sh;
    if ((e = gpu::dset(p.v.ldirty, 0, leaves * sizeof(uint32_t), p.stream))) return e;
    if ((e = gpu::dset(p.v.vdirty, 0, (p.n_cap + 1) * sizeof(uint32_t), p.stream))) return e;
    p.serial = 1;
    p.v.serial = 1;
    p.snap.gen = p.esnap.gen = 0;
  }
  if (!snap_in_step(p, sn)) {
    if ((e = snap_full_save(p, sn))) return e;
    sn.gen = p.array_gen;
  } else {
    const uint64_t leaves = p.v.g.N >> p.v.g.sh;
    GPU_LAUNCH(p.stream, k_snap_sync_leaves, (uint32_t)std::min<uint64_t>((leaves + 255) / 256, 4096), 256, p.v.items, p.v.leafcnt, sn.v.items, sn.v.leafcnt,
               p.v.ldirty, leaves, p.v.g.sh, sn.synced, 0u, 0u, (unsigned long long *)nullptr);
    if (p.v.g.n)
      GPU_LAUNCH(p.stream, k_snap_sync_nodes, (uint32_t)std::min<uint64_t>(((uint64_t)p.v.g.n + 255) / 256, 4096), 256, p.v.nodes, sn.v.nodes, p.v.vdirty,
                 (uint64_t)p.v.g.n, sn.synced, 0u, 0u);
  }
  sn.synced = p.serial;
  advance_serial(p, 1);
  return 0;
}
// snapshot -> live state (rollback / restore()); `other` is the engine's second snapshot
static int snap_rollback(Engine::Impl &p, Engine::Impl::Snap &sn, Engine::Impl::Snap &other) {
  int e;
  if (!snap_in_step(p, sn)) {
    if ((e = snap_full_load(p, sn))) return e;
    p.array_gen++;  // the live arrays were rewritten wholesale: nothing the other snapshot knows about them holds
    sn.gen = p.array_gen;
    other.gen = 0;
    sn.synced = p.serial;
    advance_serial(p, 1);
    return 0;
  }
  const uint64_t leaves = p.v.g.N >> p.v.g.sh;
  const uint32_t newtag = p.serial + 1u;  // what the rollback writes is "written" for the other snapshot
  GPU_LAUNCH(p.stream, k_snap_sync_leaves, (uint32_t)std::min<uint64_t>((leaves + 255) / 256, 4096), 256, p.v.items, p.v.leafcnt, sn.v.items, sn.v.leafcnt,
             p.v.ldirty, leaves, p.v.g.sh, sn.synced, newtag, 1u, (unsigned long long *)nullptr);
  if (p.v.g.n)
    GPU_LAUNCH(p.stream, k_snap_sync_nodes, (uint32_t)std::min<uint64_t>(((uint64_t)p.v.g.n + 255) / 256, 4096), 256, p.v.nodes, sn.v.nodes, p.v.vdirty,
               (uint64_t)p.v.g.n, sn.synced, newtag, 1u);
  sn.synced = newtag;
  advance_serial(p, 2);
  return 0;
}

int Engine::snapshot() {
  Impl &p = *p_;
  GCHK(gpu::set_device(device_));
  GCHK(snap_commit(p, p.snap));
  GCHK(gpu::sync(p.stream));
  return PPCSR_OK;
}

int Engine::restore() {
  Impl &p = *p_;
  if (!p.snap.valid) return fail(PPCSR_EINVAL, "restore without snapshot");
  GCHK(gpu::set_device(device_));
  GCHK(snap_rollback(p, p.snap, p.esnap));
  p.stamps_clean = false;
  return PPCSR_OK;
}

// Times the whole-window rebalance pipeline (rank scan + chain table + null fill + scatter + copy-back +
// recount) on the leftmost `wlen` slots without changing the final state (a rebalance is idempotent).
int Engine::rebalance_bench(uint64_t wlen, int iters, double *ms_per_call) {
  Impl &p = *p_;
  if (wlen == 0 || wlen > p.v.g.N || (wlen & (wlen - 1)) || wlen < (uint64_t)p.v.g.logN) return fail(PPCSR_EINVAL, "bad window");
  GCHK(gpu::set_device(device_));
  // (a window that starts at slot 0 crosses a binade of the position chain per doubling — a table of ~23 segments built by one
  //  thread; every other aligned window lies in one binade: option rb_bench_upper times [N - wlen, N))
  const uint64_t wstart = p.rb_bench_upper ? p.v.g.N - wlen : 0;
  int rc = big_redistribute(wstart, wlen, true);  // warm-up (also sizes the scratch array)
  if (rc != PPCSR_OK) return rc;
  p.timer.start(p.stream);  // device time of the whole pipeline: rank scan, position table, fused scatter/fill (+ copy-back)
  for (int i = 0; i < iters; i++) {
    rc = big_redistribute(wstart, wlen, false);
    if (rc != PPCSR_OK) return rc;
  }
  p.timer.stop(p.stream);
  GCHK(gpu::sync(p.stream));
  GCHK(gpu::last_error());
  *ms_per_call = p.timer.ms() / iters;
#if defined(PPCSR_SCAN_DEBUG)
  {
    unsigned long long t[32];
    GCHK(gpu::d2h(t, p.d_total, sizeof(t), p.stream));
    GCHK(gpu::sync(p.stream));
    fprintf(stderr, "[scan] cycles: scan %llu, table %llu, order: table copy %llu, flags %llu, keys %llu, count %llu, hist scan %llu, scatter %llu (100 MHz ticks x ?)\n", t[9] - t[8], t[10] - t[9],
            t[11] - t[10], t[12] - t[11], t[13] - t[12], t[14] - t[13], t[15] - t[14], t[16] - t[15]);
  }
#endif
  return inplace_fault_check();
}

// double_list / half_list alone (PCSR.cpp:251-320): the array is doubled and halved back `iters` times; device time of the
// passes (tile sums, position table, fused scatter into the fresh array), allocations and the final synchronisation outside.
// The array ends at its original size, evenly spread.
int Engine::resize_bench(int iters, double *double_ms, double *half_ms) {
  Impl &p = *p_;
  GCHK(gpu::set_device(device_));
  if (iters < 1) return fail(PPCSR_EINVAL, "bad iteration count");
  const uint64_t N0 = p.v.g.N;
  double d = 0, h = 0;
  p.time_resize = true;
  int rc = PPCSR_OK;
  for (int i = 0; i < iters && rc == PPCSR_OK; i++) {
    rc = resize(2 * N0);
    d += p.last_resize_ms;
    if (rc != PPCSR_OK) break;
    rc = resize(N0);
    h += p.last_resize_ms;
  }
  p.time_resize = false;
  if (rc != PPCSR_OK) return rc;
  *double_ms = d / iters;
  *half_ms = h / iters;
  return PPCSR_OK;
}

int bucket_ops_device(const uint32_t *starts, uint32_t n_parts, const Op *d_ops, uint64_t n, Op *d_out, unsigned long long *d_counts,
                      void *stream, std::string *errmsg) {
  static thread_local uint32_t *d_hist = nullptr;  // per-thread scratch (one rank = one process = one device)
  static thread_local uint64_t hist_cap = 0;
  const uint64_t ntiles = (n + kBucketTile - 1) / kBucketTile;
  const uint64_t need = std::max<uint64_t>(ntiles, 1) * n_parts;
  gpu::stream_t st = gpu::stream_from_ptr(stream);
  auto failm = [&](const char *what, int e) {
    if (errmsg) *errmsg = std::string(what) + ": " + gpu::err_str(e);
    return (int)PPCSR_EHIP;
  };
  if (hist_cap < need) {
    if (d_hist) gpu::dfree(d_hist);
    d_hist = nullptr;
    hist_cap = 0;
    int e = gpu::dmalloc((void **)&d_hist, need * sizeof(uint32_t));
    if (e) return failm("hipMalloc", e);
    hist_cap = need;
  }
  PartTable tab;
  for (uint32_t k = 0; k < kMaxParts; k++) tab.start[k] = k < n_parts ? starts[k] : 0xFFFFFFFFu;
  if (n == 0) {
    int e = gpu::dset(d_counts, 0, n_parts * sizeof(unsigned long long), st);
    return e ? failm("hipMemsetAsync", e) : (int)PPCSR_OK;
  }
  GPU_LAUNCH(st, k_bucket_hist, ntiles, 256, d_ops, n, tab, n_parts, d_hist);
  GPU_LAUNCH(st, k_bucket_scan, 1, kBucketScanThreads, d_hist, ntiles, n_parts, d_counts);
  GPU_LAUNCH(st, k_bucket_scatter, ntiles, 256, d_ops, n, tab, n_parts, (const uint32_t *)d_hist, d_out);
  int e = gpu::last_error();
  return e ? failm("bucket kernels", e) : (int)PPCSR_OK;
}

}  // namespace ppcsr

// ---- native RCCL exchange (include/ppcsr.h: pppcsr_comm_*, pppcsr_exchange_apply) --------------------------------------
// RCCL is bound at run time (dlopen): the engine library itself has no link dependency on it, and a process that already
// holds an RCCL (PyTorch's) shares that instance through the SONAME.
#if !defined(PPCSR_SIM)
#include <dlfcn.h>
#include <rccl/rccl.h>
namespace {
struct Rccl {
  void *lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok = false;
};
Rccl &rccl() {
  static Rccl r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (r.lib) {
#define PPCSR_SYM(f) r.f = reinterpret_cast<decltype(r.f)>(dlsym(r.lib, "nccl" #f))
      PPCSR_SYM(GetUniqueId); PPCSR_SYM(CommInitRank); PPCSR_SYM(CommDestroy); PPCSR_SYM(GroupStart); PPCSR_SYM(GroupEnd);
      PPCSR_SYM(Send); PPCSR_SYM(Recv); PPCSR_SYM(GetErrorString);
#undef PPCSR_SYM
      r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv;
    }
  }
  return r;
}
}  // namespace
struct ppcsr_xchg {  // one communicator and the stream its transfers run on
  ncclComm_t comm = nullptr;
  int nranks = 0, rank = 0, device = 0;
  gpu::stream_t stream{};
  std::string err;
};
int capi_xchg_unique_id(void *out128, std::string *err) {
  Rccl &r = rccl();
  if (!r.ok) { if (err) *err = "RCCL (librccl.so) could not be loaded"; return ppcsr::PPCSR_EHIP; }
  ncclUniqueId id;
  const ncclResult_t e = r.GetUniqueId(&id);
  if (e != ncclSuccess) { if (err) *err = std::string("ncclGetUniqueId: ") + r.GetErrorString(e); return ppcsr::PPCSR_EHIP; }
  memcpy(out128, &id, sizeof(id));
  return 0;
}
int capi_xchg_create(const void *id128, int nranks, int rank, int device, ppcsr_xchg **out, std::string *err) {
  Rccl &r = rccl();
  if (!r.ok) { if (err) *err = "RCCL (librccl.so) could not be loaded"; return ppcsr::PPCSR_EHIP; }
  if (gpu::set_device(device)) { if (err) *err = "hipSetDevice failed"; return ppcsr::PPCSR_EHIP; }
  std::unique_ptr<ppcsr_xchg> x(new ppcsr_xchg());
  x->nranks = nranks;
  x->rank = rank;
  x->device = device;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  const ncclResult_t e = r.CommInitRank(&x->comm, nranks, id, rank);
  if (e != ncclSuccess) { if (err) *err = std::string("ncclCommInitRank: ") + r.GetErrorString(e); return ppcsr::PPCSR_EHIP; }
  if (gpu::stream_create(&x->stream)) { if (err) *err = "hipStreamCreate failed"; return ppcsr::PPCSR_EHIP; }
  *out = x.release();
  return 0;
}
int capi_xchg_destroy(ppcsr_xchg *x) {
  if (!x) return 0;
  gpu::set_device(x->device);
  gpu::sync(x->stream);
  if (x->comm) rccl().CommDestroy(x->comm);
  gpu::stream_destroy(x->stream);
  delete x;
  return 0;
}
int capi_xchg_ranks(ppcsr_xchg *x, int *nranks, int *rank, int *device, void **stream) {
  if (!x) return ppcsr::PPCSR_EINVAL;
  *nranks = x->nranks;
  *rank = x->rank;
  *device = x->device;
  *stream = (void *)x->stream;
  return 0;
}
// The transport step of the owner exchange: ONE grouped set of ncclSend / ncclRecv on the communicator's stream, then a
// stream sync.  Segment i goes to / comes from peer[i]; empty segments are skipped on both sides (sender and receiver both
// know the size: the counts were exchanged first), and between one pair of ranks sends and receives match in issue order.
int capi_xchg_sendrecv(ppcsr_xchg *x, uint64_t nseg, const void *const *sptr, const uint64_t *sbytes, const int *speer, void *const *rptr,
                       const uint64_t *rbytes, const int *rpeer) {
  Rccl &r = rccl();
  auto failm = [&](const std::string &m) { x->err = m; return (int)ppcsr::PPCSR_EHIP; };
  if (gpu::set_device(x->device)) return failm("hipSetDevice failed");
  ncclResult_t e = r.GroupStart();
  for (uint64_t i = 0; i < nseg && e == ncclSuccess; i++) {
    if (sbytes[i]) e = r.Send(sptr[i], sbytes[i], ncclChar, speer[i], x->comm, x->stream);
    if (e == ncclSuccess && rbytes[i]) e = r.Recv(rptr[i], rbytes[i], ncclChar, rpeer[i], x->comm, x->stream);
  }
  const ncclResult_t e2 = r.GroupEnd();
  if (e != ncclSuccess || e2 != ncclSuccess) return failm(std::string("RCCL send/recv: ") + r.GetErrorString(e != ncclSuccess ? e : e2));
  if (gpu::sync(x->stream)) return failm("exchange stream failed");
  if (gpu::last_error()) return failm("exchange transfers failed");
  return 0;
}
const char *capi_xchg_error(ppcsr_xchg *x) { return x ? x->err.c_str() : ""; }
#endif

int gpu_device_count_for_capi(int *n) { return gpu::device_count(n); }
int capi_set_device(int d) { return gpu::set_device(d) ? ppcsr::PPCSR_EHIP : 0; }
int capi_dev_alloc(void **p, size_t bytes) { return gpu::dmalloc(p, bytes); }
int capi_dev_free(void *p) { return gpu::dfree(p); }
int capi_dev_memset(void *p, int byte, size_t bytes, void *stream) { return gpu::dset(p, byte, bytes, gpu::stream_from_ptr(stream)); }
int capi_d2h_sync(void *dst, const void *src, size_t bytes) {
  gpu::stream_t st = gpu::stream_from_ptr(nullptr);
  int e = gpu::d2h(dst, src, bytes, st);
  if (e) return e;
  return gpu::sync(st);
}
int capi_h2d_sync(void *dst, const void *src, size_t bytes) {
  gpu::stream_t st = gpu::stream_from_ptr(nullptr);
  int e = gpu::h2d(dst, src, bytes, st);
  if (e) return e;
  return gpu::sync(st);
}
