// Wave-level device routines of the PMA engine: one 64-lane wavefront executes one update.
//
// All control flow in these routines is wave-uniform (every decision is derived from ballots,
// broadcasts or wave-uniform loads), lanes cooperate on the data: a wave-wide load covers 64
// consecutive 12-byte slots (768 contiguous bytes), null/occupied masks come from __ballot and
// ranks from popcounts of the ballot mask, and the rebalance scatter is staged through LDS.
//
// Reference semantics restated here (paths relative to /root/reference/src/pcsr/):
//   pma_search        PCSR.cpp:427-502   gap-aware binary search, probe order mid, mid+1, mid-1, ...
//   plan_insert       PCSR.cpp:949-1134  acquire_insert_locks: window from PRE-insert densities,
//                                        min_node/tries bookkeeping decides normal vs global path
//   plan_remove       PCSR.cpp:1147-1232 + 597-630
//   apply_*           PCSR.cpp:519-595 (insert), 326-355 (slide_right), 597-630 (remove)
//   redistribute_wave PCSR.cpp:222-249 + fix_sentinel 168-183
#pragma once
#include "pma_geometry.h"
#include "pma_wave.h"

namespace ppcsr {

struct View {
  Edge *items;
  Node *nodes;
  uint32_t *leafcnt;
  unsigned long long *wres;  // per-leaf write reservation key of the current round
  unsigned long long *rres;  // per-leaf read reservation key (optimistic mode)
  unsigned long long *dres;  // per-leaf: earliest pending DUPLICATE update (overwrites one slot's value, moves nothing)
  unsigned long long *vw;    // per-vertex: earliest pending update that may MOVE this vertex's sentinel
  unsigned long long *vr;    // per-vertex: earliest pending update that READS this sentinel's position
  // dirty tags (incremental snapshots): every writer stamps the leaves / node records it modifies with the engine's current
  // `serial`; a snapshot that was last synchronised at serial S copies (either way) exactly the entries whose tag is > S
  uint32_t *ldirty;  // per leaf
  uint32_t *vdirty;  // per vertex (node record: beginning, end, num_neighbors)
  uint32_t serial;
  Geometry g;
  // largest rebalance window a round accepts (larger ones make the update exclusive).  Strict rounds: kBigWindow (one
  // wave rebalances it); speculative rounds: up to kBigLeaves leaves, rebalanced by a workgroup (o_big)
  uint32_t big_window;
};

constexpr uint32_t kBigWindow = 4096;  // default of View::big_window: the largest window ONE wave rebalances inside a round
constexpr uint32_t kMaxSlide = 4096;   // slides longer than this too
constexpr int kStatShards = 256;
constexpr uint32_t kLdsWindow = 256;   // windows up to this many slots are rebalanced inside one wave's LDS tile (3 KB; round 4: 512
                                       // slots = 6 KB per wave held o_apply at 6 waves per SIMD — larger windows go to o_big's workgroups)

struct StatShard {
  unsigned long long redistribute_calls, redistribute_slots, not_found, duplicates, noops, slide_slots, committed;
};

namespace dev {

PMA_DEV Edge load_stream(const Edge *p) {  // (see wv::load_stream_u32)
  const uint32_t *w = reinterpret_cast<const uint32_t *>(p);
  Edge e;
  e.src = wv::load_stream_u32(w);
  e.dest = wv::load_stream_u32(w + 1);
  e.value = wv::load_stream_u32(w + 2);
  return e;
}
PMA_DEV uint32_t lanemask_lt_count(uint64_t m, int lane) { return (uint32_t)wv::popc64(m & ((1ull << lane) - 1ull)); }

// sum of leafcnt[leaf_lo .. leaf_lo+nleaves)
PMA_DEV uint32_t count_leaves(const View &v, uint32_t leaf_lo, uint32_t nleaves) {
  if (nleaves == 1) return v.leafcnt[leaf_lo];
  uint32_t s = 0;
  // (nleaves is wave-uniform: a scalar trip count, the tail masked by a select instead of a per-lane loop)
#pragma unroll 1  // (batching these loads by unrolling costs the planning kernel 6 -> 4 waves per SIMD: 76 -> 104 VGPRs)
  for (uint32_t base = 0; base < nleaves; base += 64) {
    const uint32_t i = base + (uint32_t)wv::lane();
    const uint32_t cnt = v.leafcnt[leaf_lo + (i < nleaves ? i : nleaves - 1u)];
    s += (i < nleaves) ? cnt : 0u;
  }
  return wv::reduce_add(s);
}
// the same for the exclusive executor, whose climbs go all the way to the root (2^19 leaves at 2^24 slots: one load per trip
// made that a 3 ms walk): 16 leaves per lane per trip, four 16-byte loads in flight.  Not for the round planner — its
// climbs stop at big_window, and these registers would cost it a wave per SIMD.
PMA_DEV uint32_t count_leaves_wide(const View &v, uint32_t leaf_lo, uint32_t nleaves) {
  if (nleaves < 1024u || (leaf_lo & 3u)) return count_leaves(v, leaf_lo, nleaves);
  const uint4 *p4 = reinterpret_cast<const uint4 *>(v.leafcnt + leaf_lo);
  const uint32_t n4 = nleaves >> 2;  // (a power of two >= 256)
  uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (uint32_t i = (uint32_t)wv::lane(); i < n4; i += 256u) {
    const uint4 a = p4[i], b = p4[i + 64u], c = p4[i + 128u], d = p4[i + 192u];
    s0 += a.x + a.y + a.z + a.w;
    s1 += b.x + b.y + b.z + b.w;
    s2 += c.x + c.y + c.z + c.w;
    s3 += d.x + d.y + d.z + d.w;
  }
  return wv::reduce_add(s0 + s1 + s2 + s3);
}
template <bool WIDE>
PMA_DEV uint32_t count_window_t(const View &v, uint64_t start, uint64_t len) {
  if (WIDE) return count_leaves_wide(v, (uint32_t)(start >> v.g.sh), (uint32_t)(len >> v.g.sh));
  return count_leaves(v, (uint32_t)(start >> v.g.sh), (uint32_t)(len >> v.g.sh));
}
PMA_DEV uint32_t count_window(const View &v, uint64_t start, uint64_t len) { return count_window_t<false>(v, start, len); }

// fix the node index after sentinel `e` has been placed at slot `in` (fix_sentinel, PCSR.cpp:168-183)
PMA_DEV void fix_sentinel(const View &v, const Edge &e, uint32_t in) {
  if (!is_sentinel(e)) return;
  uint32_t vid = e.value;
  if (vid == kMax) {
    vid = 0;
  } else {
    v.nodes[vid - 1].end = in;
    v.vdirty[vid - 1] = v.serial;
  }
  v.nodes[vid].beginning = in;
  v.vdirty[vid] = v.serial;
  if (vid == v.g.n - 1) v.nodes[vid].end = (uint32_t)(v.g.N - 1);
}
// dirty tags of the leaves [lo, hi] (whole wave)
PMA_DEV void mark_leaves(const View &v, uint64_t lo, uint64_t hi) {
  for (uint64_t leaf = lo + (uint64_t)wv::lane(); leaf <= hi; leaf += 64) v.ldirty[leaf] = v.serial;
}

constexpr uint32_t kLongRange = 8;  // read ranges spanning more leaves are walked by the whole wave
// Read-leaf ranges of the update being planned.  They live in REGISTERS while the plan is made — lane r holds range r — and go to
// the plan record in one wave-wide store at the end (store_ranges): recording a range is two selects, no branch, no store.  (One
// lane-0 store per range was a divergent branch and an address computation apiece, a dozen times per plan.)
struct RangeRec {
  bool on = true;   // false: the caller wants no read set (single updates outside the rounds)
  uint32_t nr = 0;  // ranges recorded (wave-uniform), <= kMaxR = 64
  uint32_t nlong = 0;
  uint32_t my_lo = 1, my_hi = 0;
  // which sentinel positions the search result depends on: bit 0 = nodes[src].beginning (the final bracket still starts
  // at the first slot of the range), bit 1 = nodes[src].end (it still ends at the end of the range)
  uint32_t sdep = 0;
};
static_assert(kMaxR == 64, "one read range per lane");
// (slot_lo, slot_hi: wave-uniform)
PMA_DEV void rec_range(RangeRec &rr, const View &v, uint32_t slot_lo, uint32_t slot_hi) {
  if (!rr.on) return;
  const uint32_t lo = slot_lo >> v.g.sh, hi = slot_hi >> v.g.sh;
  if (rr.nr < (uint32_t)kMaxR) {
    const bool mine = (uint32_t)wv::lane() == rr.nr;
    rr.my_lo = mine ? lo : rr.my_lo;
    rr.my_hi = mine ? hi : rr.my_hi;
    rr.nlong += (hi - lo >= kLongRange) ? 1u : 0u;
    rr.nr++;
  } else {  // overflow: widen the last range (conservative)
    const bool mine = (uint32_t)wv::lane() == (uint32_t)kMaxR - 1u;
    rr.my_lo = (mine && lo < rr.my_lo) ? lo : rr.my_lo;
    rr.my_hi = (mine && hi > rr.my_hi) ? hi : rr.my_hi;
    rr.nlong++;
  }
}
// range r of the list, as a scalar (r wave-uniform)
PMA_DEV PlanRange range_at(const RangeRec &rr, uint32_t r) { return PlanRange{wv::bcast(rr.my_lo, (int)r), wv::bcast(rr.my_hi, (int)r)}; }
PMA_DEV void store_ranges(const RangeRec &rr, Plan *plan) {
  if ((uint32_t)wv::lane() < rr.nr) plan->r[wv::lane()] = PlanRange{rr.my_lo, rr.my_hi};
}

// Gap-aware lower bound of `dest` in slots [start,end) (PCSR.cpp:427-502), one wave; dest / start / end are wave-uniform and
// so is everything derived from them: the walk itself runs on scalars.
// The reference probes mid, mid+1, mid-1, mid+2, ... until it meets a live slot.  Once the interval fits one wave (<= 64
// slots) it is loaded ONCE, lane l holding slot cbase + l, and "the first live slot in probe order" is bit arithmetic on the
// ballot of the live lanes (nearest set bit right of mid against nearest set bit at or left of it; at equal distance the
// right one is probed first) — no memory round trips, no shuffles.  Longer intervals (only without the narrowing: unsorted
// ranges) probe memory, lane p taking the p-th slot of the sequence.
// *hit: what the search already knows about the slot it returns (saves the caller a dependent load): known = 1 with
// value/dest of that slot, or known = 0.
struct SearchHit {
  uint32_t known, value, dest;
  // what the walk's register copy also knows: slots [cbase, cbase + cn) were loaded (cn = 0: no copy), bit l of cnull: slot
  // cbase + l is null.  The copy reaches up to kGapAhead slots past the end of the searched interval: the gap search of an
  // insert into an occupied slot starts there and almost always ends there.
  uint32_t cbase, cn;
  uint64_t cnull;
};
constexpr uint32_t kGapAhead = 16;
// Leaf counts around the slot the search is about to return, requested WITH the walk's register copy (they were a dependent
// round trip after it): lane l < kLeafCache holds leafcnt[base + l], base a multiple of 8 — the leaf itself, its sibling of the
// "would become full" rule and the first three levels of the density climb (aligned windows of up to 8 leaves) are answered
// from registers.
constexpr uint32_t kLeafCache = 16;
struct LeafCache {
  uint32_t base = 0, n = 0;  // wave-uniform; n = 0: nothing cached
  uint32_t lc = 0;           // lane l: leafcnt[base + l]
};
PMA_DEV void leaf_cache_load(const View &v, LeafCache &lc, uint32_t slot) {
  const uint32_t nleaves = (uint32_t)(v.g.N >> v.g.sh);
  lc.base = (slot >> v.g.sh) & ~7u;
  lc.n = nleaves - lc.base < kLeafCache ? nleaves - lc.base : kLeafCache;
  if ((uint32_t)wv::lane() < lc.n) lc.lc = v.leafcnt[lc.base + (uint32_t)wv::lane()];
}
// sum of leafcnt[leaf_lo .. leaf_lo + nleaves), from the cache when it covers the range
template <bool WIDE>
PMA_DEV uint32_t count_leaves_c(const View &v, const LeafCache *lc, uint32_t leaf_lo, uint32_t nleaves) {
  if (lc != nullptr && leaf_lo >= lc->base && leaf_lo + nleaves <= lc->base + lc->n) {
    const uint32_t o = leaf_lo - lc->base;
    if (nleaves == 1) return wv::bcast(lc->lc, (int)o);
    const uint32_t l = (uint32_t)wv::lane();
    return wv::reduce_add((l >= o && l < o + nleaves) ? lc->lc : 0u);
  }
  return WIDE ? count_leaves_wide(v, leaf_lo, nleaves) : count_leaves(v, leaf_lo, nleaves);
}
template <bool WIDE>
PMA_DEV uint32_t count_window_c(const View &v, const LeafCache *lc, uint64_t start, uint64_t len) {
  return count_leaves_c<WIDE>(v, lc, (uint32_t)(start >> v.g.sh), (uint32_t)(len >> v.g.sh));
}
// What is recorded as READ is the search's CERTIFICATE, not its path.  In a sorted neighbourhood the slot the reference's
// walk returns is a function of the final tight bracket alone (tests/test_search_model.py): a live slot a with dest < key
// (or the first slot of the range), a live slot b with dest > key (or the end of the range), and nothing live in between —
// or the slot that holds the key itself.  Whatever an earlier update does to the leaves the walk merely passed through
// (the 64-ary samples, the probes of earlier iterations) cannot change the outcome as long as those bracket leaves are
// untouched, so only [a, b] is recorded; and the result depends on the POSITION of sentinel src / src + 1 only when the
// bracket still starts / ends at the range's own boundary (rr.sdep).  Path reads used to order every update of a hub vertex
// behind every write to the few leaves its coarse samples sit on.
#define PMA_CERT_BRACKET()                                                                  \
  do {                                                                                      \
    if (start == range_start) rr.sdep |= 1u;                                                \
    if (end == range_end) rr.sdep |= 2u;                                                    \
    const uint32_t _hi = (end == range_end && end > start) ? end - 1u : end;                \
    rec_range(rr, v, start < _hi ? start : _hi, start < _hi ? _hi : start);                 \
  } while (0)
// The reference's walk itself (mid, mid+1, mid-1, ... until a live slot; PCSR.cpp:427-502) on the bracket [start, end) of the
// range [range_start, range_end): the general form — unsorted / inverted ranges, brackets the 64-ary narrowing could not
// tighten (sparse samples), brackets of a single slot.  pma_search below handles the common case without a loop.
PMA_DEV uint32_t pma_search_walk(const View &v, uint32_t dest, uint32_t start, uint32_t end, uint32_t range_start, uint32_t range_end, bool narrow,
                                 RangeRec &rr, SearchHit *hit, LeafCache *lcache) {
  const int lane = wv::lane();
  const Edge *items = v.items;
  bool cached = false;
  uint32_t cbase = 0, cend = 0, cval = 0, cdst = 0;
  uint64_t clive = 0;  // bit l: slot cbase + l is live
  while (start + 1 < end) {
    if (!cached && end - start <= 64) {
      cbase = start;
      cend = end;
      const uint32_t s = start + (uint32_t)lane;
      // (the copy reaches kGapAhead slots past the interval — see SearchHit — and never past the array)
      uint64_t lim64 = (uint64_t)end + kGapAhead;
      if (lim64 > (uint64_t)start + 64ull) lim64 = (uint64_t)start + 64ull;
      if (lim64 > v.g.N) lim64 = v.g.N;
      const uint32_t lim = (uint32_t)lim64;
      if (s < lim) {
        cval = items[s].value;
        cdst = items[s].dest;
      }
      if (lcache != nullptr) leaf_cache_load(v, *lcache, start);
      clive = wv::ballot(s < end && cval != 0);
      hit->cbase = cbase;
      hit->cn = lim - cbase;
      hit->cnull = wv::ballot(s < lim && cval == 0);
      cached = true;
      if (narrow) {
        // one more narrowing step, on the register copy: the bracket comes out tight (or the key is found), and the walk
        // below ends in its first iteration
        const uint64_t mge = wv::ballot(s < end && cval != 0 && cdst >= dest);
        const uint64_t mlt = clive & ~mge;
        if (mge) {
          const int lb = wv::ctz64(mge);
          const uint32_t bdst = wv::bcast(cdst, lb);
          if (bdst == dest) {
            rec_range(rr, v, cbase + (uint32_t)lb, cbase + (uint32_t)lb);
            hit->known = 1;
            hit->value = wv::bcast(cval, lb);
            hit->dest = bdst;
            return cbase + (uint32_t)lb;
          }
          end = cbase + (uint32_t)lb;
        }
        if (mlt) start = cbase + (uint32_t)(63 - wv::clz64(mlt));
        continue;
      }
    }
    const uint32_t mid = (start + end) / 2;
    bool found = false;
    uint32_t check = mid, idest = 0, ival = 0;
    if (cached) {
      // even probes: mid - d for d = 0 .. mid - start; odd probes: mid + d for d = 1 .. end - mid - 1; probe 2d - 1 (right)
      // comes before probe 2d (left)
      const uint32_t s0 = start - cbase, m0 = mid - cbase, e0 = end - cbase;  // s0 <= m0 < e0 <= 64
      const uint64_t upto_m = (m0 >= 63u) ? ~0ull : ((2ull << m0) - 1ull);           // bits 0 .. m0
      const uint64_t below_s = (1ull << s0) - 1ull;                                    // bits 0 .. s0 - 1   (s0 <= 62)
      const uint64_t below_e = (e0 >= 64u) ? ~0ull : ((1ull << e0) - 1ull);          // bits 0 .. e0 - 1
      const uint64_t left = clive & upto_m & ~below_s, right = clive & ~upto_m & below_e;
      if (left | right) {
        found = true;
        const uint32_t lpos = left ? (uint32_t)(63 - wv::clz64(left)) : 0u, rpos = right ? (uint32_t)wv::ctz64(right) : 0u;
        const bool take_left = left && (!right || (m0 - lpos) < (rpos - m0));
        const uint32_t at = take_left ? lpos : rpos;
        check = cbase + at;
        idest = wv::bcast(cdst, (int)at);
        ival = wv::bcast(cval, (int)at);
      }
    } else {
      for (uint32_t pbase = 0;; pbase += 64) {
        const uint32_t p = pbase + (uint32_t)lane;
        const uint32_t d = (p + 1) >> 1;
        bool valid;
        uint32_t slot;
        if (p & 1u) {
          slot = mid + d;
          valid = (d < end - mid);  // mid + d < end
        } else {
          slot = mid - d;
          valid = (d <= mid - start);  // mid - d >= start
        }
        uint32_t val = 0, dst = 0;
        if (valid) {
          val = items[slot].value;
          dst = items[slot].dest;
        }
        const uint64_t m = wv::ballot(valid && val != 0);
        if (m) {
          const int pl = wv::ctz64(m);
          check = wv::bcast(slot, pl);
          idest = wv::bcast(dst, pl);
          ival = wv::bcast(val, pl);
          found = true;
          break;
        }
        if (wv::ballot(valid) == 0) break;  // both sides exhausted: the whole range is null
      }
    }
    if (!found || check == start) {
      // nothing live in (start, end): the bracket is tight
      if (found && dest <= idest) {
        if (dest == idest) rec_range(rr, v, check, check); else PMA_CERT_BRACKET();
        hit->known = 1;
        hit->value = ival;
        hit->dest = idest;
        return check;
      }
      // mid itself: the nearest live slot to it is `check` (or there is none), so unless mid == check it is null
      PMA_CERT_BRACKET();
      hit->known = 1;
      if (found && check == mid) {
        hit->value = ival;
        hit->dest = idest;
      }
      return mid;
    }
    if (dest == idest) {
      rec_range(rr, v, check, check);
      hit->known = 1;
      hit->value = ival;
      hit->dest = idest;
      return check;
    }
    if (dest < idest) end = check; else start = check;
  }
  if (end < start) start = end;
  uint32_t ev, ed;
  if (cached && start >= cbase && start < cend) {
    ev = wv::bcast(cval, (int)(start - cbase));
    ed = wv::bcast(cdst, (int)(start - cbase));
  } else {
    ev = items[start].value;
    ed = items[start].dest;
  }
  if (ev != 0 && dest == ed) {
    rec_range(rr, v, start, start);
  } else {
    PMA_CERT_BRACKET();
  }
  if (ev != 0 && dest <= ed) {
    hit->known = 1;
    hit->value = ev;
    hit->dest = ed;
    return start;
  }
  if (cached && end >= cbase && end < cbase + hit->cn) {
    hit->known = 1;
    hit->value = wv::bcast(cval, (int)(end - cbase));
    hit->dest = wv::bcast(cdst, (int)(end - cbase));
  }
  return end;
}

// The common case of pma_search: a bracket of 2 .. 64 slots in a sorted range.  true: *res is the slot the walk returns; false:
// the general walk must finish (start / end hold the bracket reached so far).
PMA_DEV bool pma_search_fast(const View &v, uint32_t dest, uint32_t &start, uint32_t &end, uint32_t range_start, uint32_t range_end, RangeRec &rr,
                             SearchHit *hit, LeafCache *lcache, uint32_t *res) {
  const int lane = wv::lane();
  const Edge *items = v.items;
  // ---- the common case: a bracket of 2 .. 64 slots in a sorted range.  It is loaded ONCE, lane l holding slot cbase + l (with
  // the leaf counts around it and kGapAhead slots behind it for the caller); one more narrowing step on that copy makes the
  // bracket tight or finds the key; the walk's first iteration on a tight bracket always ends it.  No loop.
  const uint32_t cbase = start;
  uint32_t cval = 0, cdst = 0;
  const uint32_t s = start + (uint32_t)lane;
  uint64_t lim64 = (uint64_t)end + kGapAhead;
  if (lim64 > (uint64_t)start + 64ull) lim64 = (uint64_t)start + 64ull;
  if (lim64 > v.g.N) lim64 = v.g.N;
  const uint32_t lim = (uint32_t)lim64;
  if (s < lim) {
    cval = items[s].value;
    cdst = items[s].dest;
  }
  if (lcache != nullptr) leaf_cache_load(v, *lcache, start);
  const uint64_t clive = wv::ballot(s < end && cval != 0);  // bit l: slot cbase + l is live
  hit->cbase = cbase;
  hit->cn = lim - cbase;
  hit->cnull = wv::ballot(s < lim && cval == 0);
  {
    const uint64_t mge = wv::ballot(s < end && cval != 0 && cdst >= dest);
    const uint64_t mlt = clive & ~mge;
    if (mge) {
      const int lb = wv::ctz64(mge);
      const uint32_t bdst = wv::bcast(cdst, lb);
      if (bdst == dest) {
        rec_range(rr, v, cbase + (uint32_t)lb, cbase + (uint32_t)lb);
        hit->known = 1;
        hit->value = wv::bcast(cval, lb);
        hit->dest = bdst;
        { *res = cbase + (uint32_t)lb; return true; }
      }
      end = cbase + (uint32_t)lb;
    }
    if (mlt) start = cbase + (uint32_t)(63 - wv::clz64(mlt));
  }
  if (start + 1u < end) {
    // the walk's iteration on [start, end): the first live slot in the order mid, mid+1, mid-1, ... is the nearest set bit right
    // of mid against the nearest at or left of it (at equal distance the right one is probed first)
    const uint32_t mid = (start + end) / 2;
    const uint32_t s0 = start - cbase, m0 = mid - cbase, e0 = end - cbase;  // s0 <= m0 < e0 <= 64
    const uint64_t upto_m = (m0 >= 63u) ? ~0ull : ((2ull << m0) - 1ull);           // bits 0 .. m0
    const uint64_t below_s = (1ull << s0) - 1ull;                                    // bits 0 .. s0 - 1   (s0 <= 62)
    const uint64_t below_e = (e0 >= 64u) ? ~0ull : ((1ull << e0) - 1ull);          // bits 0 .. e0 - 1
    const uint64_t left = clive & upto_m & ~below_s, right = clive & ~upto_m & below_e;
    const bool found = (left | right) != 0;
    const uint32_t lpos = left ? (uint32_t)(63 - wv::clz64(left)) : 0u, rpos = right ? (uint32_t)wv::ctz64(right) : 0u;
    const bool take_left = left && (!right || (m0 - lpos) < (rpos - m0));
    const uint32_t at = take_left ? lpos : rpos;
    const uint32_t check = found ? cbase + at : mid;
    const uint32_t idest = found ? wv::bcast(cdst, (int)at) : 0u, ival = found ? wv::bcast(cval, (int)at) : 0u;
    if (found && check != start) return false;  // (cannot happen on a tight bracket; the general walk copes if it ever does)
    if (found && dest <= idest) {
      if (dest == idest) rec_range(rr, v, check, check); else PMA_CERT_BRACKET();
      hit->known = 1;
      hit->value = ival;
      hit->dest = idest;
      { *res = check; return true; }
    }
    // nothing live in (start, end): mid is null (mid > start, and start is the only slot that can be live)
    PMA_CERT_BRACKET();
    hit->known = 1;
    { *res = mid; return true; }
  }
  // start + 1 >= end after the tightening: the walk's closing test of `start`, then `end`
  {
    if (end < start) start = end;
    const uint32_t ev = wv::bcast(cval, (int)(start - cbase)), ed = wv::bcast(cdst, (int)(start - cbase));
    if (ev != 0 && dest == ed) {
      rec_range(rr, v, start, start);
    } else {
      PMA_CERT_BRACKET();
    }
    if (ev != 0 && dest <= ed) {
      hit->known = 1;
      hit->value = ev;
      hit->dest = ed;
      { *res = start; return true; }
    }
    if (end < cbase + hit->cn) {
      hit->known = 1;
      hit->value = wv::bcast(cval, (int)(end - cbase));
      hit->dest = wv::bcast(cdst, (int)(end - cbase));
    }
    { *res = end; return true; }
  }
}

PMA_DEV uint32_t pma_search(const View &v, uint32_t dest, uint32_t start, uint32_t end, RangeRec &rr, SearchHit *hit, LeafCache *lcache = nullptr) {
  hit->known = 0;
  hit->value = 0;
  hit->dest = 0;
  hit->cbase = 0;
  hit->cn = 0;
  hit->cnull = 0;
  const int lane = wv::lane();
  const Edge *items = v.items;
  const uint32_t range_start = start, range_end = end;
  const bool narrow = v.g.narrow != 0;
  bool walk = false;  // the general walk finishes the search (one call site: it is a long piece of code)
  if (!narrow || end <= start) {  // unsorted / inverted ranges (add_node after a doubling can leave a vertex whose recorded range is
                                  // inverted, PCSR.cpp:533-540 + 681-703): no theorem, the literal walk
    rr.sdep |= 3u;
    walk = true;
  }
  // 64-ary narrowing.  The value the reference's walk returns does not depend on the walk: any bracket (start, end)
  // with "start is the range's first slot or a live slot whose dest < key" and "end is the range's end or a live slot
  // whose dest > key" leads to the same answer (the walk only ever tightens such a bracket, and every exit fires on
  // the tight one; tests/test_search_model.py checks this against the scalar walk).  So the bracket is first tightened
  // 64 samples at a time — one round trip per factor of ~64 instead of one per factor of 2 — and the walk then finishes
  // it from registers.  (One exit condition, the outcome in `state`: loops with returns inside cost the scalar unit dearly.)
  uint32_t state = 0;  // 1: a sample holds the key (fslot / fval), 2: sparse samples — no progress, the literal walk takes over
  uint32_t fslot = 0, fval = 0;
  while (!walk && state == 0u && end - start > 64u) {
    const uint32_t len = end - start;
    // samples sit on an ABSOLUTE power-of-two grid (multiples of 2^sshift), not at offsets from `start`
    const uint32_t sshift = 26u - (uint32_t)__builtin_clz(len);  // smallest shift with (len >> shift) < 64  (len > 64)
    const uint32_t first = ((start + (1u << sshift) - 1u) >> sshift) << sshift;
    const uint32_t sl = first + ((uint32_t)lane << sshift);
    uint32_t sv = 0, sd = 0;
    if (sl < end) {
      sv = items[sl].value;
      sd = items[sl].dest;
    }
    const bool live = sl < end && sv != 0;
    const uint64_t mlt = wv::ballot(live && sd < dest);
    const uint64_t mge = wv::ballot(live && sd >= dest);
    const int lb = mge ? wv::ctz64(mge) : 0;
    const uint32_t bslot = first + ((uint32_t)lb << sshift), bdst = wv::bcast(sd, lb);
    const bool key = mge != 0 && bdst == dest;  // the key itself: the walk returns its slot whenever it meets it
    fslot = bslot;
    fval = wv::bcast(sv, lb);
    const uint32_t nend = mge ? bslot : end;
    const uint32_t nstart = mlt ? first + ((uint32_t)(63 - wv::clz64(mlt)) << sshift) : start;
    const bool progress = (nend - nstart) <= len / 2u;
    if (!key) {
      start = nstart;
      end = nend;
    }
    state = key ? 1u : (progress ? 0u : 2u);  // sparse samples: let the reference's walk take over from here
  }
  if (state == 1u) {
    rec_range(rr, v, fslot, fslot);
    hit->known = 1;
    hit->value = fval;
    hit->dest = dest;
    return fslot;
  }
  if (!walk && state == 0u && start + 1u < end) {
    uint32_t res = 0;
    if (pma_search_fast(v, dest, start, end, range_start, range_end, rr, hit, lcache, &res)) return res;
  }
  return pma_search_walk(v, dest, start, end, range_start, range_end, narrow, rr, hit, lcache);
#undef PMA_CERT_BRACKET
}

// first null slot in [from, N); returns N if none.  Stops (returns kMax) after `limit` slots.
// pre / pre_nul / pre_w: the caller has already loaded the first pre_w slots (lane l < pre_w: slot from + l is null)
// pre_mask / pre_w: the caller already knows the first pre_w slots (bit l of pre_mask: slot from + l is null) — a gap among
// them costs no load at all
PMA_DEV uint32_t find_gap_right(const View &v, uint32_t from, uint32_t limit, uint64_t pre_mask = 0, uint32_t pre_w = 0) {
  const int lane = wv::lane();
  const uint64_t N = v.g.N;
  if (pre_w) {
    const uint64_t m = pre_w >= 64u ? pre_mask : (pre_mask & ((1ull << pre_w) - 1ull));
    if (m) return from + (uint32_t)wv::ctz64(m);
  }
  for (uint64_t base = (uint64_t)from + pre_w; base < N; base += 64) {
    if (base - from > limit) return kMax;
    const uint64_t s = base + (uint64_t)lane;
    bool nul = false;
    if (s < N) nul = (v.items[s].value == 0);
    const uint64_t m = wv::ballot(nul);
    if (m) return (uint32_t)(base + (uint64_t)wv::ctz64(m));
  }
  return (uint32_t)v.g.N;
}

enum PlanStatus : int { PS_OK = 0, PS_GLOBAL_NOINFO = 1, PS_GLOBAL_DOUBLE = 2, PS_SLIDE_OFF_END = 3, PS_SLIDE_LONG = 4 };
struct InsertPlan {
  int status;
  uint64_t node_index_final;
  uint64_t max_len;
  uint32_t gap;  // valid when occupied
};

// Emulation of acquire_insert_locks (PCSR.cpp:949-1134) for a single sequential caller.  The lock
// range [min_node,max_node] is tracked only as far as it steers `tries`.
// c_leaf: leafcnt of index's leaf, gap_right: find_gap_right(index + 1) when occupied — both loaded by the caller in one
// batch so that they do not sit one behind the other on the dependent-load chain.
// cap: a window beyond this many slots is not for the caller to handle (the round planner: View::big_window — the update
// turns exclusive whatever the final window is, and the exclusive executor plans again with no cap): the climb stops there
// instead of counting on towards the root, which is a walk over every leaf of the array by ONE wave (1 ms at 2^24 slots)
// while the rest of the launch waits.
template <bool WIDE = false>
PMA_DEV InsertPlan plan_insert(const View &v, uint32_t index, bool occupied, uint32_t c_leaf, uint32_t gap_right, RangeRec &rr,
                               uint64_t cap = ~0ull, const LeafCache *lcache = nullptr) {
  const Geometry &g = v.g;
  const int sh = g.sh;
  const uint64_t logN = (uint64_t)g.logN;
  InsertPlan out;
  out.status = PS_OK;
  out.node_index_final = 0;
  out.max_len = logN;
  out.gap = index;
  int64_t left_bound = -1;
  int tries = 0;
  for (;;) {
    if (tries > 3) {  // PCSR.cpp:952-955
      out.status = PS_GLOBAL_NOINFO;
      return out;
    }
    uint64_t node_index = ((uint64_t)index >> sh) << sh;
    int level = g.H;
    uint64_t len = logN;
    const int64_t node_id = (int64_t)(node_index >> sh);
    int64_t min_node = node_id;
    if (left_bound != -1) {
      if (left_bound < min_node) min_node = left_bound;
    } else if (node_id > 0 && !g.lock_search) {
      min_node = node_id - 1;
    }
    if ((uint64_t)index == g.N - 1 && occupied) {  // PCSR.cpp:992-997
      out.status = PS_GLOBAL_NOINFO;
      return out;
    }
    bool restart = false;
    // every leaf count consulted below is a READ of that leaf: record it, so that an earlier update of the same
    // round writing there is seen as a conflict even when the leaf ends up outside this update's own window
    uint32_t c = c_leaf;
    rec_range(rr, v, (uint32_t)node_index, (uint32_t)node_index);
    if ((uint64_t)c + 1 == len) {  // leaf would become full (PCSR.cpp:1012-1023)
      const uint64_t new_idx = node_index & ~(2 * len - 1);
      const int64_t new_id = (int64_t)(new_idx >> sh);
      if (new_idx != node_index && new_id < min_node) {
        left_bound = new_id;
        tries++;
        continue;
      }
      node_index = new_idx;
      c = count_leaves_c<false>(v, lcache, (uint32_t)(node_index >> sh), 1u);
      rec_range(rr, v, (uint32_t)node_index, (uint32_t)node_index);
    }
    while ((uint64_t)c + 1 >= (uint64_t)g.t_up[level]) {  // PCSR.cpp:1028-1061
      len *= 2;
      if (len > cap && len <= g.N) {
        out.max_len = len;
        out.node_index_final = node_index & ~(len - 1);
        return out;
      }
      if (len <= g.N) {
        level--;
        const uint64_t new_idx = node_index & ~(len - 1);
        if (new_idx < node_index) {
          const int64_t new_id = (int64_t)(new_idx >> sh);
          if (new_id < min_node) {
            left_bound = new_id;
            tries++;
            restart = true;
            break;
          }
          // window grew to the left: new count = old window + left half
          c += count_window_c<WIDE>(v, lcache, new_idx, len / 2);
          rec_range(rr, v, (uint32_t)new_idx, (uint32_t)(new_idx + len / 2 - 1));
          node_index = new_idx;
        } else {
          c += count_window_c<WIDE>(v, lcache, new_idx + len / 2, len / 2);
          rec_range(rr, v, (uint32_t)(new_idx + len / 2), (uint32_t)(new_idx + len - 1));
        }
      } else {
        out.status = PS_GLOBAL_DOUBLE;
        return out;
      }
    }
    if (restart) continue;
    out.max_len = len;
    out.node_index_final = node_index;
    // leaves the slide will cross (PCSR.cpp:1085-1132)
    if (occupied) {
      const uint32_t gap = gap_right;
      if (gap == kMax) {
        out.status = PS_SLIDE_LONG;
        return out;
      }
      if ((uint64_t)gap == g.N) {
        out.status = PS_SLIDE_OFF_END;
        return out;
      }
      out.gap = gap;
    }
    return out;
  }
}

// The first attempt of plan_insert (tries = 0, no left bound yet), for the round planner: 32-bit arithmetic (its windows stop at
// `cap` <= 2^16 slots), straight-line but for the climb, whose one exit carries the outcome in `st`.  false: the attempt ended
// in the reference's retry path (the window wants to grow to the left of the leaves it "holds", PCSR.cpp:1016-1022 /
// 1043-1049) — the caller restores its read set and runs the general plan_insert.  Everything else: *out as plan_insert
// would leave it.
PMA_DEV bool plan_insert_first(const View &v, uint32_t index, bool occupied, uint32_t c_leaf, uint32_t gap_right, RangeRec &rr, uint32_t cap,
                               const LeafCache *lcache, InsertPlan *out) {
  const Geometry &g = v.g;
  const int sh = g.sh;
  const uint32_t logN = (uint32_t)g.logN;
  const uint32_t N32 = g.N > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)g.N;  // (window lengths here stay <= 2 * cap)
  out->status = PS_OK;
  out->node_index_final = 0;
  out->max_len = logN;
  out->gap = index;
  if ((uint64_t)index == g.N - 1 && occupied) {  // PCSR.cpp:992-997
    out->status = PS_GLOBAL_NOINFO;
    return true;
  }
  uint32_t node_index = (index >> sh) << sh;
  const uint32_t node_id = index >> sh;
  const uint32_t min_node = (node_id > 0u && !g.lock_search) ? node_id - 1u : node_id;
  uint32_t len = logN;
  int level = g.H;
  uint32_t c = c_leaf;
  rec_range(rr, v, node_index, node_index);
  if (c + 1u == len) {  // leaf would become full (PCSR.cpp:1012-1023)
    const uint32_t new_idx = node_index & ~(2u * len - 1u);
    if (new_idx != node_index && (new_idx >> sh) < min_node) return false;
    node_index = new_idx;
    c = count_leaves_c<false>(v, lcache, node_index >> sh, 1u);
    rec_range(rr, v, node_index, node_index);
  }
  uint32_t st = 0;  // 1: window beyond cap, 2: retry path, 3: root overflow
  while (st == 0u && c + 1u >= g.t_up[level]) {  // PCSR.cpp:1028-1061
    len *= 2u;
    if (len > N32) {
      st = 3u;
    } else if (len > cap) {
      st = 1u;
    } else {
      level--;
      const uint32_t new_idx = node_index & ~(len - 1u);
      const bool left = new_idx < node_index;
      if (left && (new_idx >> sh) < min_node) {
        st = 2u;
      } else {
        // the window grew to the left (new count = old window + left half) or to the right
        const uint32_t half = left ? new_idx : new_idx + len / 2u;
        c += count_window_c<false>(v, lcache, half, len / 2u);
        rec_range(rr, v, half, half + len / 2u - 1u);
        node_index = new_idx;
      }
    }
  }
  if (st == 2u) return false;
  if (st == 3u) {
    out->status = PS_GLOBAL_DOUBLE;
    return true;
  }
  out->max_len = len;
  if (st == 1u) {
    out->node_index_final = node_index & ~(len - 1u);
    return true;
  }
  out->node_index_final = node_index;
  // leaves the slide will cross (PCSR.cpp:1085-1132)
  if (occupied) {
    if (gap_right == kMax) out->status = PS_SLIDE_LONG;
    else if ((uint64_t)gap_right == g.N) out->status = PS_SLIDE_OFF_END;
    else out->gap = gap_right;
  }
  return true;
}

// NOTE on the count update inside the climb above: the reference recomputes get_density over the whole
// new window at every level; the window at level L-1 is the union of the level-L window and its sibling,
// so adding the sibling's count gives the identical integer.  In the "would become full" quirk case the
// level-H window is the LEFT sibling leaf (len stays logN); the union argument still holds from there.

struct RemovePlan {
  int half;  // climb reached the root: half_list()
  uint64_t wstart, wlen;
};
template <bool WIDE = false>
PMA_DEV RemovePlan plan_remove(const View &v, uint32_t index, RangeRec &rr, uint64_t cap = ~0ull, const LeafCache *lcache = nullptr) {
  const Geometry &g = v.g;
  const int sh = g.sh;
  RemovePlan out;
  out.half = 0;
  uint64_t node_index = ((uint64_t)index >> sh) << sh;
  int level = g.H;
  uint64_t len = (uint64_t)g.logN;
  uint32_t c = count_leaves_c<false>(v, lcache, (uint32_t)(node_index >> sh), 1u);  // pre-removal count; compare c-1
  rec_range(rr, v, (uint32_t)node_index, (uint32_t)node_index);
  while ((uint64_t)c < (uint64_t)g.t_lo[level] + 1) {  // (c - 1) < t_lo
    len *= 2;
    if (len > cap && len <= g.N) break;  // (wlen > cap: exclusive, see plan_insert)
    if (len <= g.N) {
      level--;
      const uint64_t new_idx = node_index & ~(len - 1);
      if (new_idx < node_index) {
        c += count_window_c<WIDE>(v, lcache, new_idx, len / 2);
        rec_range(rr, v, (uint32_t)new_idx, (uint32_t)(new_idx + len / 2 - 1));
        node_index = new_idx;
      } else {
        c += count_window_c<WIDE>(v, lcache, new_idx + len / 2, len / 2);
        rec_range(rr, v, (uint32_t)(new_idx + len / 2), (uint32_t)(new_idx + len - 1));
      }
    } else {
      out.half = 1;
      break;
    }
  }
  out.wstart = node_index;
  out.wlen = len;
  return out;
}

// ---- in-wave window rebalance (redistribute, PCSR.cpp:222-249) -----------------------------------------
// Window of <= 64 slots: one coalesced load, ballot + popcount ranks, exact fp64 position chain,
// scatter staged through this wave's LDS tile, one coalesced store.  Larger windows (rare) are
// streamed through the same wave in 64-slot chunks: stable in-place compaction to the left, null fill,
// then spread right-to-left — the reference's own three phases, 64 slots at a time.
// Window of <= 64 slots whose slots are in registers (lane l: slot wstart + l; lanes >= wlen hold nulls): ranks, exact
// positions, scatter through the wave's LDS tile, one coalesced store, leaf counts.
PMA_DEV void redistribute_regs(const View &v, uint64_t wstart, uint64_t wlen, const Edge &e, uint32_t *lds) {
  const int lane = wv::lane();
  Edge *items = v.items;
  const int sh = v.g.sh;
  const uint32_t logN = (uint32_t)v.g.logN;
  const bool valid = (uint64_t)lane < wlen;
  const bool nn = valid && e.value != 0;
  const uint64_t m = wv::ballot(nn);
  const uint32_t j = (uint32_t)wv::popc64(m);
  const uint32_t k = lanemask_lt_count(m, lane);
  uint64_t mypos = wstart;
  if (j >= 2) {
    ChainSeg sg;
    if (chain_single(wstart, wlen, j, &sg)) {  // closed form (every window that does not start at slot 0)
      if (nn) mypos = chain_single_pos(sg, wstart, j, k);
    } else {
      const double step = chain_step(wlen, j);
      double x = chain_top(wstart, j, step);
      for (uint32_t t = 0; t + 1 < j; t++) {
        if (nn && k == j - 1 - t) mypos = (uint64_t)x;
        x = chain_sub(x, step);
      }
    }
  }
  // LDS tile (SoA, stride-1): clear, scatter, gather
  lds[lane] = kMax;
  lds[64 + lane] = 0;
  lds[128 + lane] = 0;
  wv::lds_fence();
  if (nn) {
    const uint32_t o = (uint32_t)(mypos - wstart);
    lds[o] = e.src;
    lds[64 + o] = e.dest;
    lds[128 + o] = e.value;
    fix_sentinel(v, e, (uint32_t)mypos);
  }
  wv::lds_fence();
  Edge out;
  out.src = lds[lane];
  out.dest = lds[64 + lane];
  out.value = lds[128 + lane];
  if (valid) items[wstart + lane] = out;
  const uint64_t occ = wv::ballot(valid && out.value != 0);
  const uint32_t nleaf = (uint32_t)(wlen >> sh);
  if ((uint32_t)lane < nleaf) {
    const uint64_t sub = (logN >= 64) ? occ : ((occ >> ((uint32_t)lane * logN)) & ((1ull << logN) - 1ull));
    v.leafcnt[(wstart >> sh) + lane] = (uint32_t)wv::popc64(sub);
  }
  wv::fence();
}
PMA_DEV void redistribute_wave(const View &v, uint64_t wstart, uint64_t wlen, uint32_t *lds /* 3*kLdsWindow u32 per wave */) {
  const int lane = wv::lane();
  Edge *items = v.items;
  const int sh = v.g.sh;
  const uint32_t logN = (uint32_t)v.g.logN;
  if (wlen <= 64) {
    Edge e = null_edge();
    if ((uint64_t)lane < wlen) e = items[wstart + lane];
    redistribute_regs(v, wstart, wlen, e, lds);
    return;
  }
  if (wlen <= kLdsWindow) {
    // ---- LDS-staged path (64 < wlen <= kLdsWindow): the whole window lives in this wave's LDS tile ------
    // global -> LDS (independent coalesced loads), stable in-place compaction, right-to-left spread with the
    // exact position chain (the reference's three phases, PCSR.cpp:226-247, on LDS), LDS -> global.
    uint32_t *ls = lds, *ld = lds + kLdsWindow, *lv = lds + 2 * kLdsWindow;
    const uint32_t W = (uint32_t)wlen;
    for (uint32_t o = (uint32_t)lane; o < W; o += 64) {
      const Edge e = items[wstart + o];
      ls[o] = e.src;
      ld[o] = e.dest;
      lv[o] = e.value;
    }
    wv::lds_fence();
    uint32_t j = 0;
    for (uint32_t base = 0; base < W; base += 64) {  // compaction (chunk fully read before it is rewritten)
      const uint32_t o = base + (uint32_t)lane;
      const uint32_t es = ls[o], ed = ld[o], ev = lv[o];
      const bool nn = ev != 0;
      const uint64_t m = wv::ballot(nn);
      const uint32_t k = j + lanemask_lt_count(m, lane);
      wv::lds_fence();
      if (nn) {
        ls[k] = es;
        ld[k] = ed;
        lv[k] = ev;
      }
      j += (uint32_t)wv::popc64(m);
      wv::lds_fence();
    }
    for (uint32_t o = j + (uint32_t)lane; o < W; o += 64) {
      ls[o] = kMax;
      ld[o] = 0;
      lv[o] = 0;
    }
    wv::lds_fence();
    if (j >= 2) {
      const double step = chain_step(wlen, j);
      double x = chain_top(wstart, j, step);
      ChainSeg sg;
      const bool single = chain_single(wstart, wlen, j, &sg);
      uint32_t khi = j - 1;
      while (khi >= 1) {
        const uint32_t klo = (khi >= 64) ? khi - 63 : 1;
        const uint32_t cntc = khi - klo + 1;
        uint64_t mypos = 0;
        if (single) {
          if ((uint32_t)lane < cntc) mypos = chain_single_pos(sg, wstart, j, khi - (uint32_t)lane);
        } else {
          for (uint32_t i = 0; i < cntc; i++) {
            if ((uint32_t)lane == i) mypos = (uint64_t)x;
            x = chain_sub(x, step);
          }
        }
        const bool act = (uint32_t)lane < cntc;
        const uint32_t so = khi - (uint32_t)lane;  // source offset in the compacted tile
        uint32_t es = kMax, ed = 0, ev = 0;
        if (act) {
          es = ls[so];
          ed = ld[so];
          ev = lv[so];
        }
        wv::lds_fence();
        const uint32_t po = (uint32_t)(mypos - wstart);
        if (act && po != so) {
          ls[so] = kMax;
          ld[so] = 0;
          lv[so] = 0;
        }
        wv::lds_fence();
        if (act && po != so) {
          ls[po] = es;
          ld[po] = ed;
          lv[po] = ev;
        }
        if (act) fix_sentinel(v, Edge{es, ed, ev}, (uint32_t)mypos);
        wv::lds_fence();
        khi = klo - 1;
      }
    }
    if (j >= 1 && lane == 0) fix_sentinel(v, Edge{ls[0], ld[0], lv[0]}, (uint32_t)wstart);
    wv::lds_fence();
    for (uint32_t base = 0; base < W; base += 64) {  // LDS -> global + leaf counts
      const uint32_t o = base + (uint32_t)lane;
      Edge e;
      e.src = ls[o];
      e.dest = ld[o];
      e.value = lv[o];
      items[wstart + o] = e;
      const uint64_t occ = wv::ballot(e.value != 0);
      const uint32_t nleaf = (logN >= 64) ? 1u : (64u >> sh);
      if ((uint32_t)lane < nleaf) {
        const uint64_t sub = (logN >= 64) ? occ : ((occ >> ((uint32_t)lane * logN)) & ((1ull << logN) - 1ull));
        v.leafcnt[((wstart + base) >> sh) + lane] = (uint32_t)wv::popc64(sub);
      }
    }
    wv::fence();
    return;
  }
  // ---- chunked path -----------------------------------------------------------------------------
  const uint64_t wend = wstart + wlen;
  uint64_t wr = wstart;
  for (uint64_t base = wstart; base < wend; base += 64) {  // phase 1: stable compaction to the left
    const Edge e = items[base + lane];
    const bool nn = e.value != 0;
    const uint64_t m = wv::ballot(nn);
    const uint32_t k = lanemask_lt_count(m, lane);
    wv::fence();
    if (nn) items[wr + k] = e;
    wr += (uint64_t)wv::popc64(m);
    wv::fence();
  }
  const uint64_t j = wr - wstart;
  for (uint64_t s = wstart + j + (uint64_t)lane; s < wend; s += 64) items[s] = null_edge();  // phase 1.5
  wv::fence();
  if (j >= 2) {  // phase 2: spread right-to-left, 64 elements per step
    const double step = chain_step(wlen, j);
    double x = chain_top(wstart, j, step);
    ChainSeg sg;
    const bool single = chain_single(wstart, wlen, j, &sg);
    uint64_t khi = j - 1;
    while (khi >= 1) {
      const uint64_t klo = (khi >= 64) ? khi - 63 : 1;
      const uint32_t cntc = (uint32_t)(khi - klo + 1);
      uint64_t mypos = 0;
      if (single) {
        if ((uint32_t)lane < cntc) mypos = chain_single_pos(sg, wstart, j, khi - (uint64_t)lane);
      } else {
        for (uint32_t i = 0; i < cntc; i++) {
          if ((uint32_t)lane == i) mypos = (uint64_t)x;
          x = chain_sub(x, step);
        }
      }
      const bool act = (uint32_t)lane < cntc;
      const uint64_t srcslot = wstart + (khi - (uint64_t)lane);
      Edge e = null_edge();
      if (act) e = items[srcslot];
      wv::fence();
      if (act && mypos != srcslot) items[srcslot] = null_edge();
      wv::fence();
      if (act && mypos != srcslot) items[mypos] = e;
      if (act) fix_sentinel(v, e, (uint32_t)mypos);
      wv::fence();
      khi = klo - 1;
    }
  }
  if (j >= 1 && lane == 0) {
    const Edge e0 = items[wstart];
    fix_sentinel(v, e0, (uint32_t)wstart);
  }
  wv::fence();
  // phase 3: recount the window's leaves
  for (uint64_t base = wstart; base < wend; base += 64) {
    const Edge e = items[base + lane];
    const uint64_t occ = wv::ballot(e.value != 0);
    const uint32_t nleaf = (logN >= 64) ? 1u : (64u >> sh);
    if ((uint32_t)lane < nleaf) {
      const uint64_t sub = (logN >= 64) ? occ : ((occ >> ((uint32_t)lane * logN)) & ((1ull << logN) - 1ull));
      v.leafcnt[(base >> sh) + lane] = (uint32_t)wv::popc64(sub);
    }
  }
  wv::fence();
}

// ---- workgroup window rebalance (redistribute, PCSR.cpp:222-249, for windows of up to kBigLeaves leaves) --------------
// One workgroup, out of place through a scratch stretch of its own, two streaming passes with no dependency between
// chunks: (1) every wave takes 64-slot chunks, ranks their live elements (leaf prefix from the exact leaf counts, kept in
// LDS, + ballot/popcount inside the leaf), places each at its exact position and fills the null run up to the next
// element's position — every scratch slot is written exactly once; (2) the scratch stretch is copied back and the leaf
// counts are rewritten from ballots.  Sentinels are fixed up in pass 1 (fix_sentinel).
constexpr uint32_t kBigLeaves = 4096;    // leaves per job at most (131072 slots at logN = 32)
constexpr uint32_t kBigThreads = 1024;
struct BigShared {
  uint32_t pref[kBigLeaves + 1];
  uint32_t wsum[kBigThreads / 64];
  ChainTable tb;
};
PMA_DEV void redistribute_block(const View &v, uint64_t wstart, uint64_t wlen, Edge *scratch, BigShared &sh) {
  const int lane = wv::lane(), w = wv::wave_in_block();
  const uint32_t tid = wv::thread_idx(), nthreads = wv::block_dim(), nwaves = nthreads >> 6;
  const int s = v.g.sh;
  const uint32_t logN = (uint32_t)v.g.logN;
  const uint32_t nleaf = (uint32_t)(wlen >> s);
  const uint32_t lf0 = (uint32_t)(wstart >> s);
  Edge *items = v.items;
  // exclusive prefix of the leaf counts: every thread owns a run of `per` consecutive leaves
  const uint32_t per = (nleaf + nthreads - 1) / nthreads;
  uint32_t mine = 0;
  for (uint32_t q = 0; q < per; q++) {
    const uint32_t i = tid * per + q;
    if (i < nleaf) mine += v.leafcnt[lf0 + i];
  }
  uint32_t incl = mine;
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = wv::shfl(incl, lane - o < 0 ? 0 : lane - o);
    if (lane >= o) incl += y;
  }
  if (lane == 63) sh.wsum[w] = incl;
  wv::block_sync();
  uint32_t woff = 0, j = 0;
  for (uint32_t q = 0; q < nwaves; q++) {
    const uint32_t x = sh.wsum[q];
    if (q < (uint32_t)w) woff += x;
    j += x;
  }
  {
    uint32_t run = woff + incl - mine;
    for (uint32_t q = 0; q < per; q++) {
      const uint32_t i = tid * per + q;
      if (i < nleaf) {
        sh.pref[i] = run;
        run += v.leafcnt[lf0 + i];
      }
    }
  }
  ChainSeg sg;
  const bool single = chain_single(wstart, wlen, j, &sg);  // every aligned window that does not start at slot 0
  if (!single && j >= 2 && tid == 0) build_chain_table(wstart, wlen, j, &sh.tb);
  wv::block_sync();
  const uint64_t wend = wstart + wlen;
  const uint32_t nchunks = (uint32_t)(wlen >> 6);
  if (j == 0) {
    for (uint64_t o = tid; o < wlen; o += nthreads) scratch[o] = null_edge();
  } else {
    int hint = -1, hint2 = -1;
    constexpr int kB = 4;  // chunks requested back to back per wave
    for (uint32_t c0 = (uint32_t)w * kB; c0 < nchunks; c0 += nwaves * kB) {
      Edge e[kB];
#pragma unroll
      for (int b = 0; b < kB; b++) {
        e[b] = null_edge();
        if (c0 + b < nchunks) e[b] = items[wstart + (uint64_t)(c0 + b) * 64 + lane];
      }
#pragma unroll
      for (int b = 0; b < kB; b++) {
        if (c0 + b >= nchunks) break;
        const uint32_t off = (c0 + b) * 64u + (uint32_t)lane;
        const bool nn = e[b].value != 0;
        const uint64_t m = wv::ballot(nn);
        if (m == 0) continue;
        if (nn) {
          const uint32_t first = (logN >= 64) ? 0u : ((uint32_t)lane & ~(logN - 1u));
          const uint64_t lmask = (logN >= 64) ? ~0ull : (((1ull << logN) - 1ull) << first);
          const uint64_t k = (uint64_t)sh.pref[off >> s] + (uint64_t)wv::popc64(m & lmask & ((1ull << lane) - 1ull));
          uint64_t pos, nxt;
          if (single) {
            pos = chain_single_pos(sg, wstart, j, k);
            nxt = (k + 1 < j) ? chain_single_pos(sg, wstart, j, k + 1) : wend;
          } else if (j >= 2) {
            pos = chain_pos(&sh.tb, k, &hint);
            nxt = (k + 1 < j) ? chain_pos(&sh.tb, k + 1, &hint2) : wend;
          } else {
            pos = wstart;
            nxt = wend;
          }
          scratch[pos - wstart] = e[b];
          for (uint64_t q = pos + 1; q < nxt; q++) scratch[q - wstart] = null_edge();
          fix_sentinel(v, e[b], (uint32_t)pos);
        }
      }
    }
  }
  wv::block_sync();  // (workgroup-scope release/acquire: every wave sees the others' scratch stores)
  const uint32_t lpc = (logN >= 64) ? 1u : (64u >> s);  // leaves per 64-slot chunk
  for (uint32_t c = (uint32_t)w; c < nchunks; c += nwaves) {
    const Edge e = scratch[(uint64_t)c * 64 + lane];
    items[wstart + (uint64_t)c * 64 + lane] = e;
    const uint64_t occ = wv::ballot(e.value != 0);
    if ((uint32_t)lane < lpc) {
      const uint64_t sub = (logN >= 64) ? occ : ((occ >> ((uint32_t)lane * logN)) & ((1ull << logN) - 1ull));
      v.leafcnt[((wstart + (uint64_t)c * 64) >> s) + lane] = (uint32_t)wv::popc64(sub);
    }
  }
}

// shift items[index .. gap-1] one slot to the right (slide_right, PCSR.cpp:326-355); gap is null.
PMA_DEV void slide_right_wave(const View &v, uint32_t index, uint32_t gap) {
  const int lane = wv::lane();
  Edge *items = v.items;
  uint64_t hi = gap;  // exclusive end of the run still to move
  while (hi > index) {
    const uint64_t lo = (hi - index > 64) ? hi - 64 : index;
    const uint64_t s = lo + (uint64_t)lane;
    const bool act = s < hi;
    Edge e = null_edge();
    if (act) e = items[s];
    wv::fence();
    if (act) {
      items[s + 1] = e;
      fix_sentinel(v, e, (uint32_t)(s + 1));
    }
    wv::fence();
    hi = lo;
  }
}

// last null slot in [0, from]; kMax if there is none (slide_left's scan, PCSR.cpp:367)
PMA_DEV uint32_t find_gap_left(const View &v, uint32_t from) {
  const int lane = wv::lane();
  for (int64_t hi = (int64_t)from; hi >= 0; hi -= 64) {
    const int64_t s = hi - (int64_t)lane;
    bool nul = false;
    if (s >= 0) nul = (v.items[s].value == 0);
    const uint64_t m = wv::ballot(nul);
    if (m) return (uint32_t)(hi - (int64_t)wv::ctz64(m));
  }
  return kMax;
}
// shift items[gap+1 .. last] one slot to the left (the net effect of slide_left(last), PCSR.cpp:360-390, whose carry
// stops at the null slot `gap`); slot `last` is left for the caller to overwrite
PMA_DEV void slide_left_wave(const View &v, uint32_t gap, uint32_t last) {
  const int lane = wv::lane();
  Edge *items = v.items;
  for (uint64_t lo = (uint64_t)gap + 1; lo <= (uint64_t)last; lo += 64) {
    const uint64_t s = lo + (uint64_t)lane;
    const bool act = s <= (uint64_t)last;
    Edge e = null_edge();
    if (act) e = items[s];
    wv::fence();
    if (act) {
      items[s - 1] = e;
      fix_sentinel(v, e, (uint32_t)(s - 1));
    }
    wv::fence();
  }
}

// ---- full per-op planning (search + window plan) -------------------------------------------------------
// What the planning kernels need from the plan right away (the full record goes to memory for the later kernels).  Everything
// but my_lo / my_hi is wave-uniform (scalar registers).
struct PlanRegs {
  uint32_t kind, index, wstart, wlen, wleaf_lo, wleaf_hi, mv_lo, mv_hi, nr, nlong, sdep, sleaf_b, sleaf_e;
  uint32_t my_lo, my_hi;  // lane r: read range r (r < 64)
};
// the header words of a plan record, lane i holding word i: one wave-wide store (the record's first 80 bytes)
PMA_DEV void store_plan_header(Plan *plan, uint32_t kind, uint32_t index, uint32_t gap, uint32_t wstart, uint32_t wlen, uint32_t wl, uint32_t wh,
                               uint32_t mv_lo, uint32_t mv_hi, uint32_t sleaf_b, uint32_t sleaf_e, uint32_t acalls, uint32_t aslots, uint32_t nr,
                               uint32_t nlong, uint32_t sdep, uint32_t idx, const Op &op) {
  uint32_t w = 0;
  w = wv::setlane<PW_KIND>(w, kind);
  w = wv::setlane<PW_INDEX>(w, index);
  w = wv::setlane<PW_GAP>(w, gap);
  w = wv::setlane<PW_WSTART>(w, wstart);
  w = wv::setlane<PW_WLEN>(w, wlen);
  w = wv::setlane<PW_WLEAF_LO>(w, wl);
  w = wv::setlane<PW_WLEAF_HI>(w, wh);
  w = wv::setlane<PW_MV_LO>(w, mv_lo);
  w = wv::setlane<PW_MV_HI>(w, mv_hi);
  w = wv::setlane<PW_SLEAF_B>(w, sleaf_b);
  w = wv::setlane<PW_SLEAF_E>(w, sleaf_e);
  w = wv::setlane<PW_ALG>(w, (acalls << 28) | aslots);
  w = wv::setlane<PW_NR>(w, nr);
  w = wv::setlane<PW_NLONG>(w, nlong);
  w = wv::setlane<PW_SDEP>(w, sdep);
  w = wv::setlane<PW_IDX>(w, idx);
  w = wv::setlane<PW_SRC>(w, op.src);
  w = wv::setlane<PW_DST>(w, op.dst);
  w = wv::setlane<PW_OP>(w, op.op);
  if (wv::lane() < (int)PW_HEADER_WORDS) reinterpret_cast<uint32_t *>(plan)[wv::lane()] = w;
}
// op, idx (the update's stream index): wave-uniform
PMA_DEV PlanRegs plan_op(const View &v, const Op op, Plan *plan, uint32_t idx) {
  const int lane = wv::lane();
  const Geometry &g = v.g;
  RangeRec rr;
  uint32_t kind = K_NOOP, index = 0, gap = 0, wstart = 0, wlen = 0, wl = 1, wh = 0, acalls = 0, aslots = 0;
  uint32_t sleaf_b = 0, sleaf_e = 0, mv_lo = 1, mv_hi = 0;
  // The node records around `src`, requested in ONE batch before anything depends on them: lanes 0 .. kW-1 hold the sentinel
  // positions of src, src-1, ... (lane 0 also nodes[src].end), lanes kW .. 2kW-1 those of src+1, src+2, ... — the vertex' own
  // range now, the sentinels inside the update's write range later (the loads do not depend on the window, only the
  // comparisons do: asked for after the window plan they were one more round trip at the end of every plan)
  constexpr uint32_t kW = 8;
  uint32_t b0 = 0, e0 = 0;
  bool v0 = false;
  if (op.src < g.n) {
    // (one predicated load for all 2 kW lanes: {beginning, end} as an 8-byte access — only lane 0's `end` is used)
    const bool dn = (uint32_t)lane < kW;
    const uint64_t u = dn ? (uint64_t)op.src - (uint64_t)lane : (uint64_t)op.src + 1ull + ((uint64_t)lane - kW);
    v0 = (uint32_t)lane < 2u * kW && (dn ? (uint32_t)lane <= op.src : u < g.n);
    if (v0) {
      const Node *nd = &v.nodes[u];
      b0 = nd->beginning;
      e0 = nd->end;
    }
  }
  if (op.src < g.n) {
    // nodes[src].{beginning,end} are the positions of sentinels src / src+1: that dependency is tracked per vertex
    // (Plan::mv_lo/mv_hi of the writers, View::vw/vr), not through the leaves that hold them
    const uint32_t nd_beginning = wv::bcast(b0, 0), nd_end = wv::bcast(e0, 0);
    sleaf_b = nd_beginning >> g.sh;
    sleaf_e = nd_end >> g.sh;
    SearchHit hit;
    LeafCache lcache;
    index = pma_search(v, op.dst, nd_beginning + 1, nd_end, rr, &hit, &lcache);
    const uint32_t leaf = index >> g.sh;
    // What is still missing, in one batch of independent loads: the slot the search returned (unless the search already
    // knows it), the leaf counts around it (unless they came with the walk's register copy) and — for an insert the copy
    // does not reach behind — the first slots of the gap search to the right
    Edge at;
    at.src = 0;
    at.value = hit.value;
    at.dest = hit.dest;
    if (!hit.known) {
      at.value = v.items[index].value;
      at.dest = v.items[index].dest;
    }
    if (!(lcache.n && leaf >= lcache.base && leaf < lcache.base + lcache.n)) leaf_cache_load(v, lcache, index);
    // slots right of `index` the register copy holds: bit l of gmask = slot index + 1 + l is null, for l < gw
    uint64_t gmask = 0;
    uint32_t gw = 0;
    if (hit.cn && index >= hit.cbase && index + 1u < hit.cbase + hit.cn) {
      const uint32_t sft = index + 1u - hit.cbase;  // 1 .. 63
      gmask = hit.cnull >> sft;
      gw = hit.cn - sft;
    }
    constexpr uint32_t kGapPre = 16;  // slots of the gap search requested with this batch (a null within 16 slots in 99.7 % of the cases at density 0.7)
    if (op.op != 0 && gw == 0u) {
      const uint64_t g0 = (uint64_t)index + 1ull + (uint64_t)lane;
      bool nul0 = false;
      if ((uint32_t)lane < kGapPre && g0 < g.N) nul0 = (v.items[g0].value == 0);
      gmask = wv::ballot(nul0);
      const uint64_t room = g.N - 1ull - (uint64_t)index;  // slots right of index
      gw = room < kGapPre ? (uint32_t)room : kGapPre;
    }
    const uint32_t c_leaf = wv::bcast(lcache.lc, (int)(leaf - lcache.base));
    const bool occupied = !is_null(at);
    if (op.op != 0) {
      const Edge elem{op.src, op.dst, op.op};
      if (occupied && !is_sentinel(elem) && at.dest == op.dst) {
        kind = K_DUP;
        wl = wh = leaf;
      } else {
        const uint32_t gap_right = occupied ? find_gap_right(v, index + 1, kMaxSlide, gmask, gw) : index;
        InsertPlan ip;
        {
          const RangeRec rr0 = rr;
          if (!plan_insert_first(v, index, occupied, c_leaf, gap_right, rr, v.big_window, &lcache, &ip)) {
            rr = rr0;  // (the general form starts over, retries included)
            ip = plan_insert(v, index, occupied, c_leaf, gap_right, rr, v.big_window, &lcache);
          }
        }
        // tries > 3 (PCSR.cpp:952-955): the reference gives up on leaf locks, takes the global write lock and runs
        // insert(..., nullptr) — same slide, same write, but the window comes from POST-insert densities (PCSR.cpp:578-590).
        // That climb is a function of the leaf counts and of where the slide's gap is, so it is planned here like any
        // other window instead of sending the update to the exclusive executor (in a structure whose every level sits
        // close to its density bound almost every multi-level climb ends this way).  Slot N-1 occupied
        // (PCSR.cpp:992-997: double, re-search) stays exclusive.
        bool noinfo = false;
        if (ip.status == PS_GLOBAL_NOINFO && !((uint64_t)index == g.N - 1 && occupied) && g.logN <= 32 &&
            !(occupied && (gap_right == kMax || (uint64_t)gap_right == g.N))) {
          noinfo = true;
          ip.status = PS_OK;
          ip.gap = occupied ? gap_right : index;
          ip.max_len = (uint64_t)g.logN;
        }
        if (ip.status != PS_OK) {
          kind = K_EXCL;
        } else {
          gap = ip.gap;
          const uint32_t gleaf = gap >> g.sh;
          const uint32_t cpost = c_leaf + ((gleaf == leaf) ? 1u : 0u);
          uint64_t ws, wn;
          if (cpost == (uint32_t)g.logN) {  // PCSR.cpp:555-557
            wn = 2ull * (uint64_t)g.logN;
            ws = ((uint64_t)index) & ~(wn - 1);
          } else {
            wn = (uint64_t)g.logN;
            ws = (uint64_t)leaf << g.sh;
          }
          acalls = 1;
          aslots = (uint32_t)wn;
          bool need_double = false;
          if (noinfo && ws + wn <= g.N) {
            // the first density the reference looks at is that of (node_index, logN) AFTER the leaf / 2-leaf pass
            // (PCSR.cpp:555-564): with a 2-leaf pass that is the evened-out LEFT leaf
            rec_range(rr, v, (uint32_t)ws, (uint32_t)(ws + wn - 1));
            uint32_t c = cpost, j2 = 0;
            if (wn != (uint64_t)g.logN) {
              const uint32_t l0 = (uint32_t)(ws >> g.sh);
              j2 = count_leaves_c<false>(v, &lcache, l0, 2u) + ((gleaf == l0 || gleaf == l0 + 1u) ? 1u : 0u);
              // elements of the evened window that land in its left leaf: the literal position chain (PCSR.cpp:237-247),
              // at most 63 dependent subtractions, every lane the same (cheap in registers: this path is rare and must
              // not cost the planning kernel its occupancy)
              c = (j2 >= 1) ? 1u : 0u;  // element 0 stays on the window's first slot
              if (j2 >= 2) {
                const double step = chain_step(wn, j2);
                double x = chain_top(ws, j2, step);
                const uint64_t mid = ws + (uint64_t)g.logN;
                for (uint32_t t = 0; t + 1 < j2; t++) {
                  c += ((uint64_t)x < mid) ? 1u : 0u;
                  x = chain_sub(x, step);
                }
              }
            }
            uint64_t node_index = ws, len = (uint64_t)g.logN;
            int level = g.H;
            while ((uint64_t)c >= (uint64_t)g.t_up[level]) {
              len *= 2;
              if (len > g.N) {
                need_double = true;
                break;
              }
              if (len > v.big_window) break;  // exclusive whatever the final window is (see plan_insert's cap)
              level--;
              const uint64_t new_idx = node_index & ~(len - 1);
              if (len == wn) {
                c = j2;  // the 2-leaf window itself (its right leaf was evened by the same pass)
              } else {
                const uint64_t half = (new_idx < node_index) ? new_idx : new_idx + len / 2;
                c += count_window_c<false>(v, &lcache, half, len / 2) + (((uint64_t)gleaf >= (half >> g.sh) && (uint64_t)gleaf < ((half + len / 2) >> g.sh)) ? 1u : 0u);
                rec_range(rr, v, (uint32_t)half, (uint32_t)(half + len / 2 - 1));
              }
              node_index = new_idx;
            }
            if (!need_double && len > wn) {  // PCSR.cpp:592-594
              ws = node_index;
              wn = len;
              acalls = 2;
              aslots += (uint32_t)wn;
            } else if (!need_double && len > (uint64_t)g.logN) {
              acalls = 2;  // (the climb stopped at the 2-leaf window: the reference rebalances it a second time)
              aslots += (uint32_t)len;
            }
          } else if (ip.max_len > (uint64_t)g.logN) {  // PCSR.cpp:592-594
            ws = ip.node_index_final;
            wn = ip.max_len;
            acalls = 2;
            aslots += (uint32_t)wn;
          }
          if (need_double || wn > v.big_window || ws + wn > g.N) {
            kind = K_EXCL;
          } else {
            kind = K_INSERT;
            wstart = (uint32_t)ws;
            wlen = (uint32_t)wn;
            const uint32_t a = (uint32_t)(ws >> g.sh), b = (uint32_t)((ws + wn - 1) >> g.sh);
            wl = a < leaf ? a : leaf;
            wh = b > gleaf ? b : gleaf;
          }
        }
      }
    } else {
      const Edge elem{op.src, op.dst, 1u};
      if (!occupied || is_sentinel(elem) || at.dest != op.dst) {
        kind = K_NOTFOUND;
      } else {
        const RemovePlan rp = plan_remove(v, index, rr, v.big_window, &lcache);
        if (rp.half || rp.wlen > v.big_window) {
          kind = K_EXCL;
        } else {
          kind = K_REMOVE;
          acalls = 2;  // leaf pass + window pass, always both (PCSR.cpp:609, 629)
          aslots = (uint32_t)g.logN + (uint32_t)rp.wlen;
          wstart = (uint32_t)rp.wstart;
          wlen = (uint32_t)rp.wlen;
          wl = (uint32_t)(rp.wstart >> g.sh);
          wh = (uint32_t)((rp.wstart + rp.wlen - 1) >> g.sh);
        }
      }
    }
  } else if (op.op == 0) {
    kind = K_NOOP;  // reference: unchecked out-of-range delete is UB; we ignore it
  }
  if (kind == K_INSERT || kind == K_REMOVE) {
    // sentinels inside [lo, hi] = slide range U window.  Sentinel positions increase with the vertex id and
    // beg(src) < index <= beg(src+1), so they are the vertices src, src-1, ... and src+1, src+2, ... around `src`.
    // The first kW of both directions are in registers already (b0): no round trip for almost every update — a one-leaf
    // window holds a handful of sentinels at most.
    const uint32_t lo = (wstart < index) ? wstart : index;
    const uint32_t whi = wstart + wlen - 1u;
    const uint32_t hi = (kind == K_INSERT && gap > whi) ? gap : whi;
    const uint64_t mdn = wv::ballot((uint32_t)lane < kW && v0 && b0 >= lo);
    const uint64_t mup = wv::ballot((uint32_t)lane >= kW && (uint32_t)lane < 2u * kW && v0 && b0 <= hi) >> kW;
    uint32_t beg_lowest = 0;  // beginning of the lowest vertex found inside the range (vertex mv_lo)
    uint32_t down = (uint32_t)wv::popc64(mdn), up = (uint32_t)wv::popc64(mup);
    if (mdn) beg_lowest = wv::bcast(b0, 63 - wv::clz64(mdn));  // (the in-range vertices are a prefix of the lanes)
    if (mdn == ((1ull << kW) - 1ull)) {
      for (uint32_t base = kW;; base += 64) {  // downwards: src - kW, ...
        const uint32_t k = base + (uint32_t)lane;
        uint32_t bg = 0;
        bool in = false;
        if (k <= op.src) {
          bg = v.nodes[op.src - k].beginning;
          in = bg >= lo;
        }
        const uint64_t m = wv::ballot(in);
        if (m) beg_lowest = wv::bcast(bg, 63 - wv::clz64(m));
        down += (uint32_t)wv::popc64(m);
        if (m != ~0ull) break;
      }
    }
    if (mup == ((1ull << kW) - 1ull)) {
      for (uint32_t base = kW;; base += 64) {  // upwards: src + 1 + kW, ...
        const uint64_t u = (uint64_t)op.src + 1ull + base + (uint64_t)lane;
        bool in = false;
        if (u < g.n) in = v.nodes[u].beginning <= hi;
        const uint64_t m = wv::ballot(in);
        up += (uint32_t)wv::popc64(m);
        if (m != ~0ull) break;
      }
    }
    mv_lo = op.src + 1u - down;
    mv_hi = op.src + up;
    // the element at the first slot of the rebalance window keeps its slot (PCSR.cpp:240: "already in the correct
    // position"); a sentinel sitting there does not move unless the insert/slide displaces it
    if (mv_lo <= mv_hi) {
      // (down == 0: the lowest vertex inside is src + 1, whose sentinel position is lane kW's)
      const uint32_t b_lo = (down > 0) ? beg_lowest : wv::bcast(b0, (int)kW);
      if (b_lo == wstart && (kind == K_REMOVE || index > wstart)) mv_lo++;
    }
  }
  const uint32_t nr = rr.nr;
  store_ranges(rr, plan);
  store_plan_header(plan, kind, index, gap, wstart, wlen, wl, wh, mv_lo, mv_hi, sleaf_b, sleaf_e, acalls, aslots, nr, rr.nlong, rr.sdep, idx, op);
  PlanRegs pr;
  pr.sdep = rr.sdep;
  pr.kind = kind;
  pr.index = index;
  pr.wstart = wstart;
  pr.sleaf_b = sleaf_b;
  pr.sleaf_e = sleaf_e;
  pr.wlen = wlen;
  pr.wleaf_lo = wl;
  pr.wleaf_hi = wh;
  pr.mv_lo = mv_lo;
  pr.mv_hi = mv_hi;
  pr.nr = nr;
  pr.nlong = rr.nlong;
  pr.my_lo = rr.my_lo;
  pr.my_hi = rr.my_hi;
  return pr;
}

// The plan record's header and this lane's read range, requested in ONE batch: lane i loads header word i, lane r < kHeadRanges
// range r — two wave-wide loads inside the record's first 128 bytes — and the fields are read off the lanes as scalars.  The
// update itself is in there too (idx, src, dst, op): o_check / o_apply need neither the op array nor the slot's index list.
constexpr int kHeadRanges = 6;
struct PlanHead {
  uint32_t kind, index, gap, wstart, wlen, wleaf_lo, wleaf_hi, mv_lo, mv_hi, sleaf_b, sleaf_e, alg_calls, alg_slots, nr, nlong, sdep;
  uint32_t idx;  // stream index of the update
  Op op;         // the update
  uint32_t my_lo, my_hi;  // lane r: read range r (r < kHeadRanges; longer lists are walked from the record)
};
PMA_DEV PlanHead load_plan_head(const Plan *pl) {
  const int lane = wv::lane();
  uint32_t w = 0;
  PlanRange rg{1u, 0u};
  if (lane < (int)PW_HEADER_WORDS) w = reinterpret_cast<const uint32_t *>(pl)[lane];
  if (lane < kHeadRanges) rg = pl->r[lane];
  uint32_t f[PW_HEADER_WORDS];
  wv::lanes<PW_HEADER_WORDS>(w, f);
  PlanHead h;
  h.kind = f[PW_KIND];
  h.index = f[PW_INDEX];
  h.gap = f[PW_GAP];
  h.wstart = f[PW_WSTART];
  h.wlen = f[PW_WLEN];
  h.wleaf_lo = f[PW_WLEAF_LO];
  h.wleaf_hi = f[PW_WLEAF_HI];
  h.mv_lo = f[PW_MV_LO];
  h.mv_hi = f[PW_MV_HI];
  h.sleaf_b = f[PW_SLEAF_B];
  h.sleaf_e = f[PW_SLEAF_E];
  h.alg_calls = f[PW_ALG] >> 28;
  h.alg_slots = f[PW_ALG] & 0x0FFFFFFFu;
  h.idx = f[PW_IDX];
  h.op = Op{f[PW_SRC], f[PW_DST], f[PW_OP]};
  h.nr = f[PW_NR];
  h.nlong = f[PW_NLONG];
  h.sdep = f[PW_SDEP];
  h.my_lo = rg.lo;
  h.my_hi = rg.hi;
  return h;
}

// A window too large for one wave is handed to a workgroup (o_big) through the round's job queue: the update's own wave
// does everything up to the final rebalance (slide, write, counters) and leaves the leaf counts exact.
struct BigJob {
  uint32_t wstart, wlen;
};
// apply a planned op whose reservations were validated.  defer != nullptr: queue the rebalance instead of running it here.
PMA_DEV void apply_op(const View &v, const Op op, const PlanHead &h, uint32_t *lds, StatShard *st, BigJob *defer = nullptr) {
  const PlanHead *plan = &h;
  const int lane = wv::lane();
  const Geometry &g = v.g;
  const uint32_t kind = plan->kind;
  const uint32_t index = plan->index;
  if (kind == K_NOOP) {
    if (lane == 0) wv::atomic_add_u64(&st->noops, 1ull);
    return;
  }
  if (kind == K_DUP) {
    if (lane == 0) {
      v.items[index].value = op.op;
      v.ldirty[index >> g.sh] = v.serial;
      v.vdirty[op.src] = v.serial;
      wv::atomic_add_nn(&v.nodes[op.src].num_neighbors, 1u);
      wv::atomic_add_u64(&st->duplicates, 1ull);
    }
    return;
  }
  if (kind == K_NOTFOUND) {
    if (lane == 0) {
      v.vdirty[op.src] = v.serial;
      wv::atomic_add_nn(&v.nodes[op.src].num_neighbors, 0xFFFFFFFFu);
      wv::atomic_add_u64(&st->not_found, 1ull);
    }
    return;
  }
  if (kind == K_INSERT || kind == K_REMOVE) {  // everything this update writes lies in [wleaf_lo, wleaf_hi] (slide + window)
    mark_leaves(v, plan->wleaf_lo, plan->wleaf_hi);
    if (lane == 0) v.vdirty[op.src] = v.serial;
  }
  // A window of <= 64 slots that contains everything the update touches (slot, slide) is loaded ONCE: the slide and the write
  // happen in registers, then ranks / positions / one store.  (Slide, write and rebalance one after the other were three
  // load -> store -> acknowledgement round trips on the same 64 slots; sentinels the slide moves are inside the window, whose
  // rebalance sets every back-pointer from the final positions.)
  if ((kind == K_INSERT || kind == K_REMOVE) && !defer && plan->wlen <= 64u && index >= plan->wstart &&
      (kind == K_REMOVE || plan->gap < plan->wstart + plan->wlen)) {
    const uint32_t ws = plan->wstart, wn = plan->wlen;
    Edge e = null_edge();
    if ((uint32_t)lane < wn) e = v.items[ws + (uint32_t)lane];
    const uint32_t li = index - ws;
    if (kind == K_INSERT) {
      const uint32_t lg = plan->gap - ws;  // li <= lg < wn
      if (lane == 0) {
        wv::atomic_add_nn(&v.nodes[op.src].num_neighbors, 1u);
        wv::atomic_add_u64(&st->redistribute_calls, (unsigned long long)plan->alg_calls);
        wv::atomic_add_u64(&st->redistribute_slots, (unsigned long long)plan->alg_slots);
        wv::atomic_add_u64(&st->slide_slots, (unsigned long long)(lg - li));
      }
      const int from = lane > 0 ? lane - 1 : 0;
      Edge up;
      up.src = wv::shfl(e.src, from);
      up.dest = wv::shfl(e.dest, from);
      up.value = wv::shfl(e.value, from);
      if ((uint32_t)lane > li && (uint32_t)lane <= lg) e = up;  // slide_right: slots index .. gap-1 move one to the right
      if ((uint32_t)lane == li) e = Edge{op.src, op.dst, op.op};
    } else {
      if (lane == 0) {
        wv::atomic_add_nn(&v.nodes[op.src].num_neighbors, 0xFFFFFFFFu);
        wv::atomic_add_u64(&st->redistribute_calls, (unsigned long long)plan->alg_calls);
        wv::atomic_add_u64(&st->redistribute_slots, (unsigned long long)plan->alg_slots);
      }
      if ((uint32_t)lane == li) {
        e.value = 0;
        e.dest = 0;
      }
    }
    redistribute_regs(v, ws, wn, e, lds);
    return;
  }
  if (kind == K_INSERT) {
    const uint32_t gap = plan->gap;
    if (gap != index) slide_right_wave(v, index, gap);
    // the rebalance below recounts every leaf of its window; only a gap that lies beyond the window needs its leaf's
    // count bumped here (a read-modify-write the following fence would otherwise have to wait for)
    const bool gap_outside = defer || ((uint64_t)gap < (uint64_t)plan->wstart) || ((uint64_t)gap >= (uint64_t)plan->wstart + plan->wlen);
    if (lane == 0) {
      v.items[index] = Edge{op.src, op.dst, op.op};
      if (gap_outside) v.leafcnt[gap >> g.sh] += 1u;  // (a slide moves one element over every leaf boundary it crosses: only the gap's leaf gains)
      wv::atomic_add_nn(&v.nodes[op.src].num_neighbors, 1u);
      wv::atomic_add_u64(&st->redistribute_calls, (unsigned long long)plan->alg_calls);
      wv::atomic_add_u64(&st->redistribute_slots, (unsigned long long)plan->alg_slots);
      wv::atomic_add_u64(&st->slide_slots, (unsigned long long)(gap - index));
      if (defer) *defer = BigJob{plan->wstart, plan->wlen};
    }
    wv::fence();
    if (!defer) redistribute_wave(v, plan->wstart, plan->wlen, lds);
    return;
  }
  if (kind == K_REMOVE) {
    if (lane == 0) {
      v.items[index].value = 0;
      v.items[index].dest = 0;
      if (defer || (uint64_t)index < (uint64_t)plan->wstart || (uint64_t)index >= (uint64_t)plan->wstart + plan->wlen)
        v.leafcnt[index >> g.sh] -= 1u;  // (in-wave: never — the window contains the slot and its leaves are recounted below)
      wv::atomic_add_nn(&v.nodes[op.src].num_neighbors, 0xFFFFFFFFu);
      wv::atomic_add_u64(&st->redistribute_calls, (unsigned long long)plan->alg_calls);
      wv::atomic_add_u64(&st->redistribute_slots, (unsigned long long)plan->alg_slots);
      if (defer) *defer = BigJob{plan->wstart, plan->wlen};
    }
    wv::fence();
    if (!defer) redistribute_wave(v, plan->wstart, plan->wlen, lds);
    return;
  }
}

}  // namespace dev
}  // namespace ppcsr
