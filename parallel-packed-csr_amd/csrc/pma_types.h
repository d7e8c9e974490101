// Core types of the MI355X packed-memory-array engine (shared by host and device code).
//
// HBM layout (bit-identical to the reference's in-memory layout so that state export is a
// plain copy and parity is checked byte-for-byte):
//   items[N]   : 12-byte AoS slots {src, dest, value}            (reference edge_t, PCSR.h:30-35)
//   nodes[n]   : 12-byte {beginning, end, num_neighbors}          (reference node_t, PCSR.h:18-23)
// Auxiliary device-only arrays (derived state, never exported):
//   leafcnt[N/logN] : live slots per PMA leaf   (replaces the reference's get_density rescans,
//                                                PCSR.cpp:126-133; same values, no re-reading)
//   wres/rres[N/logN] : per-leaf write/read reservations of the current scheduling round
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PMA_HD __host__ __device__
#else
#define PMA_HD
#endif

namespace ppcsr {

struct Edge {
  uint32_t src, dest, value;
};
struct Node {
  uint32_t beginning, end, num_neighbors;
};
struct Op {  // op == 0: delete (src,dst); op != 0: add (src,dst) with edge value `op`
  uint32_t src, dst, op;
};

constexpr uint32_t kMax = 0xFFFFFFFFu;
constexpr int kMaxLevels = 36;
constexpr uint32_t kNever = 0xFFFFFFFFu;

// Geometry of the implicit calibrator tree + integer density thresholds.
// Level L in [0,H]: window length len(L) = logN << (H - L)  (level H = leaf, level 0 = root).
//   insert climbs while  count + 1 >= t_up[L]   <=>  count/len + 1/len >= 3/4 + (.25*L)/H   (PCSR.cpp:1026-1028,163)
//   remove climbs while  count - 1 <  t_lo[L]   <=>  count/len - 1/len <  1/4 - (.125*L)/H  (PCSR.cpp:1193-1197,162)
// (densities are exact dyadic rationals, so the fp64 comparisons of the reference reduce to
//  integer comparisons against thresholds computed once on the host with the same fp64 expressions)
struct Geometry {
  uint64_t N;
  uint32_t n;
  int logN, sh, H;
  int lock_search;
  // 1: neighbourhood ranges are sorted and disjoint (always, unless add_node has hit the reference's re-search-after-
  // doubling path, PCSR.cpp:533-540, which can drop the new sentinel into the middle of another vertex's range: from
  // then on the reference searches unsorted ranges and only the literal walk follows it) -> the 64-ary narrowing is valid
  uint32_t narrow;
  uint32_t t_up[kMaxLevels];
  uint32_t t_lo[kMaxLevels];
};

PMA_HD inline bool is_null(const Edge &e) { return e.value == 0; }
PMA_HD inline bool is_sentinel(const Edge &e) { return e.dest == kMax || e.value == kMax; }  // PCSR.cpp:64
PMA_HD inline Edge null_edge() { return Edge{kMax, 0u, 0u}; }

// ---- per-op plan produced by the plan phase and consumed by check/apply ---------------------------
enum Kind : uint32_t {
  K_NOOP = 0,      // add with src >= n (silently ignored, PCSR.cpp:1375)
  K_INSERT = 1,    // new edge: optional slide, write, one window rebalance
  K_DUP = 2,       // edge exists: overwrite value only (PCSR.cpp:529-532)
  K_REMOVE = 3,    // edge found: null the slot, one window rebalance
  K_NOTFOUND = 4,  // delete of a missing edge: only num_neighbors-- (PCSR.cpp:747-754)
  K_EXCL = 5,      // must run alone through the exclusive executor (global path, resize, big window)
  K_SKIP = 6,      // already executed by the exclusive executor inside this epoch: commits as nothing
};

constexpr int kMaxR = 64;  // read-leaf ranges recorded per op (search certificate + one or two per level of the density climb + a few): one per lane of the planning wave

// Layout: the 20-word header and the first 6 read ranges fill the record's first 128 bytes, which is all the round kernels touch
// for almost every update (two or three ranges: the search certificate and the leaves of a short climb) — one wave-wide store
// in o_plan, one wave-wide load in o_check / o_apply (lane i <-> word i; the fields are then read off the lanes as scalars).
// The update itself (stream index, src, dst, op) travels in the header too: the later kernels of the round need nothing else.
struct PlanRange {
  uint32_t lo, hi;  // inclusive leaf range
};
struct Plan {
  uint32_t kind;
  uint32_t index;             // slot returned by the gap-aware search
  uint32_t gap;               // first null slot at or right of index (== index if items[index] is null)
  uint32_t wstart, wlen;      // the single window rebalance that the op performs
  uint32_t wleaf_lo, wleaf_hi;  // inclusive leaf range written (slide + window)
  // sentinel dependencies, tracked per VERTEX instead of through the leaf that happens to hold the sentinel:
  // the update reads the positions of sentinels `src` and `src+1` (nodes[src].beginning / .end) and may move the
  // sentinels of vertices [mv_lo, mv_hi] (those inside its slide range / rebalance window; empty if mv_lo > mv_hi)
  uint32_t mv_lo, mv_hi;
  uint32_t sleaf_b, sleaf_e;  // leaves currently holding sentinel src / src+1 (growth-zone check of deferred writers)
  uint32_t alg;               // redistribute() calls (bits 28..31) and slots (bits 0..27) the reference performs for this op
  uint32_t nr;
  uint32_t nlong;  // number of read ranges spanning >= kLongRange leaves (0 for almost every update)
  uint32_t sdep;   // bit 0 / 1: the search result depends on the position of sentinel src / src + 1 (pma_search)
  uint32_t idx;    // stream index of the update
  uint32_t src, dst, op;  // the update
  uint32_t pad;
  PlanRange r[kMaxR];  // leaf ranges read by the search / the density climb
  uint32_t pad2[12];    // (records are five 128-byte lines)
};
// word index of a header field inside the record (lane <-> word in the wave-wide accesses)
enum PlanWord : int { PW_KIND = 0, PW_INDEX, PW_GAP, PW_WSTART, PW_WLEN, PW_WLEAF_LO, PW_WLEAF_HI, PW_MV_LO, PW_MV_HI, PW_SLEAF_B, PW_SLEAF_E,
                      PW_ALG, PW_NR, PW_NLONG, PW_SDEP, PW_IDX, PW_SRC, PW_DST, PW_OP, PW_PAD, PW_HEADER_WORDS };
static_assert(sizeof(Plan) == 640 && PW_HEADER_WORDS == 20, "plan record layout");

// scheduler control block (device memory, mirrored to pinned host memory between round chunks)
struct Control {
  uint32_t base[2];      // stream position of the first pending op, double-buffered by round parity
  uint32_t horizon[2];   // ops planned this round
  uint32_t failmin[2];   // smallest stream index that failed its reservation check
  uint32_t n_ops;
  uint32_t excl;         // 1: the op at base[] needs the exclusive executor
  uint32_t error;        // sticky device-side error code
  uint32_t max_horizon;
  // statistics (monotone counters)
  unsigned long long rounds, committed, planned;
  unsigned long long redistribute_calls, redistribute_slots;  // algorithmic: what the reference would do
  unsigned long long not_found, duplicates, noops;
  unsigned long long slide_slots;
};

// result of the exclusive executor
enum ExclResult : uint32_t {
  X_DONE = 0,
  X_NEED_DOUBLE = 1,        // double_list() completes the op (PCSR.cpp:570-572, 586-588)
  X_NEED_HALF = 2,          // half_list() completes the op (PCSR.cpp:624-626)
  X_NEED_REDIST = 3,        // host must run the multi-workgroup window rebalance on (wstart,wlen)
  X_DOUBLE_THEN_RETRY = 4,  // slot N-1 occupied: double_list(), re-search, insert(..., nullptr) (PCSR.cpp:533-540)
  X_UNSUPPORTED = 5,        // a slide found no null slot on either side (reference: PCSR.cpp:378-383) — never observed
  X_VIOLATION = 7,          // speculative epoch only: a LATER update has already been committed on something this update reads
                            // or writes — the epoch must be rolled back (validation by stamps, as for every other update)
  X_WINDOW_BEYOND_ARRAY = 6,  // the 2-leaf rebalance of a full leaf on a ONE-leaf array (N == logN): the reference reads and
                              // writes past the end of its array there (PCSR.cpp:555-557 with len*2 > N) — undefined behaviour
};
struct ExclOut {
  uint32_t result;
  uint32_t wstart, wlen;
  uint32_t found;  // edge_exists / misc
};

enum ErrorCode : int {
  PPCSR_OK = 0,
  PPCSR_EINVAL = 1,
  PPCSR_ENOMEM = 2,
  PPCSR_EHIP = 3,
  PPCSR_EUNSUPPORTED = 4,
  PPCSR_EINTERNAL = 5,
  PPCSR_ERANGE = 6,
};

}  // namespace ppcsr
