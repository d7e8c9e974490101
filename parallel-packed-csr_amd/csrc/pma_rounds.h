// Kernels of the PMA engine.  Launch geometry: 256-thread workgroups = 4 wavefronts; the update
// kernels give one wavefront to one update ("wave per op"), the whole-array kernels give one
// wavefront to 64 consecutive slots (768 contiguous bytes per wave-wide access).
//
// Scheduling model (replaces the reference's per-leaf locks, PCSR.cpp:949-1232, and its thread pools):
// a batch is applied in ROUNDS.  Each round plans the next `horizon` pending updates of the stream
// against the current state (k_plan), every plan reserves the PMA leaves it would write with an
// atomicMin of its stream index, k_check finds the first update whose read or write leaves were
// reserved by an EARLIER update, and k_apply executes the conflict-free PREFIX before that update.
// Inside such a prefix no update reads or writes anything an earlier one writes, so executing them
// concurrently is identical to the reference's sequential stream order.
#pragma once
#include "pma_device.h"

// Visit every leaf of the plan's read ranges: lane r owns range r (ranges are almost always 1-2 leaves), so the
// ranges are processed side by side instead of one dependent loop iteration after another.
#define PMA_FOR_EACH_READ_LEAF(pl, lane, LEAFVAR, BODY)                                     \
  do {                                                                                      \
    const uint32_t _nr = (pl)->nr;                                                          \
    for (uint32_t _r = (uint32_t)(lane); _r < _nr; _r += 64) {                              \
      const uint32_t _lo = (pl)->r[_r].lo, _hi = (pl)->r[_r].hi;                              \
      if (_hi - _lo < dev::kLongRange)                                                      \
        for (uint32_t LEAFVAR = _lo; LEAFVAR <= _hi; LEAFVAR++) { BODY; }                   \
    }                                                                                       \
    if ((pl)->nlong) { /* rare: long ranges are walked by all lanes together */             \
      for (uint32_t _r = 0; _r < _nr; _r++) {                                               \
        const uint32_t _lo = (pl)->r[_r].lo, _hi = (pl)->r[_r].hi;                            \
        if (_hi - _lo >= dev::kLongRange)                                                   \
          for (uint32_t LEAFVAR = _lo + (uint32_t)(lane); LEAFVAR <= _hi; LEAFVAR += 64) { BODY; } \
      }                                                                                     \
    }                                                                                       \
  } while (0)

namespace ppcsr {

struct RoundArgs {
  View v;
  const Op *ops;
  Plan *plans;
  Control *ctl;
  StatShard *stats;
  uint32_t round;
  uint32_t min_horizon;
};

PMA_DEV unsigned long long make_key(uint32_t round, uint32_t idx) {
  return ((unsigned long long)(0xFFFFFFFFu - round) << 32) | (unsigned long long)idx;
}
PMA_DEV bool kind_writes(uint32_t k) { return k == K_INSERT || k == K_DUP || k == K_REMOVE; }
// A duplicate insert (K_DUP, PCSR.cpp:529-532) only overwrites the `value` of one existing slot: searches test
// value != 0 and compare `dest`, neither of which changes, so it conflicts with updates that MOVE or rewrite slots of
// that leaf (same round only; across rounds they commute) but never with readers.  Strong writers move slots.
PMA_DEV bool kind_strong(uint32_t k) { return k == K_INSERT || k == K_REMOVE; }
PMA_DEV bool kind_real(uint32_t k) { return k != K_NOOP && k != K_SKIP; }  // has a source vertex and a place in the array

PMA_KERNEL void k_plan(RoundArgs a) {
  Control *c = a.ctl;
  const uint32_t par = a.round & 1u;
  if (c->excl || c->error) return;
  if (wv::block_idx() == 0 && wv::thread_idx() == 0) c->failmin[par ^ 1u] = kMax;
  const uint32_t base = c->base[par], hor = c->horizon[par];
  const uint32_t wid = wv::uni(wv::block_idx() * (wv::block_dim() >> 6) + (uint32_t)wv::wave_in_block());
  if (wid >= hor) return;
  const uint32_t idx = base + wid;
  const Op op = a.ops[idx];
  Plan *pl = &a.plans[wid];
  const dev::PlanRegs pr = dev::plan_op(a.v, op, pl, idx);
  const uint32_t kind = pr.kind;
  if (kind == K_DUP) {
    if (wv::lane() == 0) wv::atomic_min_u64(&a.v.dres[pr.wleaf_lo], make_key(a.round, idx));
  } else if (kind_strong(kind)) {
    const unsigned long long key = make_key(a.round, idx);
    const uint32_t wl = pr.wleaf_lo, wh = pr.wleaf_hi;
    for (uint32_t leaf = wl + (uint32_t)wv::lane(); leaf <= wh; leaf += 64) wv::atomic_min_u64(&a.v.wres[leaf], key);
    const uint32_t ml = pr.mv_lo, mh = pr.mv_hi;  // sentinels this update may move
    for (uint64_t u = (uint64_t)ml + (uint64_t)wv::lane(); u <= (uint64_t)mh && ml <= mh; u += 64) wv::atomic_min_u64(&a.v.vw[u], key);
  }
}

PMA_KERNEL void k_check(RoundArgs a) {
  Control *c = a.ctl;
  const uint32_t par = a.round & 1u;
  if (c->excl || c->error) return;
  const uint32_t base = c->base[par], hor = c->horizon[par];
  const uint32_t wid = wv::uni(wv::block_idx() * (wv::block_dim() >> 6) + (uint32_t)wv::wave_in_block());
  if (wid >= hor) return;
  const uint32_t idx = base + wid;
  const Plan *pl = &a.plans[wid];
  const uint32_t kind = pl->kind;
  const unsigned long long key = make_key(a.round, idx);
  const uint32_t tag = (uint32_t)(key >> 32);
  bool fail = (kind == K_EXCL);
  if (kind == K_DUP) {
    const uint32_t leaf = pl->wleaf_lo;
    const unsigned long long kw = a.v.wres[leaf];
    if ((uint32_t)(kw >> 32) == tag && (uint32_t)kw < idx) fail = true;  // an earlier update moves slots of this leaf
    if (a.v.dres[leaf] != key) fail = true;                               // an earlier duplicate on this leaf
  } else if (kind_strong(kind)) {
    const uint32_t wl = pl->wleaf_lo, wh = pl->wleaf_hi;
    for (uint32_t leaf = wl + (uint32_t)wv::lane(); leaf <= wh; leaf += 64) {
      if (a.v.wres[leaf] != key) fail = true;  // an earlier update writes this leaf
      const unsigned long long kd = a.v.dres[leaf];
      if ((uint32_t)(kd >> 32) == tag && (uint32_t)kd < idx) fail = true;  // an earlier duplicate overwrites a slot here
    }
  }
  PMA_FOR_EACH_READ_LEAF(pl, wv::lane(), leaf, {
    const unsigned long long k = a.v.wres[leaf];
    if ((uint32_t)(k >> 32) == tag && (uint32_t)k < idx) fail = true;  // an earlier update writes what we read
  });
  if (kind != K_NOOP && pl->sdep) {  // the result depends on nodes[src].beginning / .end: an earlier update moves that sentinel
    const uint32_t src = a.ops[idx].src;
    if (src < a.v.g.n) {
      const unsigned long long k0 = a.v.vw[src];
      if ((pl->sdep & 1u) && (uint32_t)(k0 >> 32) == tag && (uint32_t)k0 < idx) fail = true;
      if ((pl->sdep & 2u) && src + 1u < a.v.g.n) {
        const unsigned long long k1 = a.v.vw[src + 1u];
        if ((uint32_t)(k1 >> 32) == tag && (uint32_t)k1 < idx) fail = true;
      }
    }
  }
  if (wv::ballot(fail) != 0 && wv::lane() == 0 && idx < c->failmin[par]) wv::atomic_min_u32(&c->failmin[par], idx);
}

PMA_KERNEL void k_apply(RoundArgs a) {
  PMA_SHARED uint32_t lds[4][3 * kLdsWindow];
  Control *c = a.ctl;
  const uint32_t par = a.round & 1u;
  if (c->error) return;
  const uint32_t base = c->base[par], hor = c->horizon[par];
  if (hor == 0) {
    if (wv::block_idx() == 0 && wv::thread_idx() == 0) {
      c->base[par ^ 1u] = base;
      c->horizon[par ^ 1u] = 0;
    }
    return;
  }
  const uint32_t fm = c->failmin[par];
  const uint32_t limit = (fm < base + hor) ? fm : base + hor;
  if (wv::block_idx() == 0 && wv::thread_idx() == 0) {
    const uint32_t committed = limit - base;
    uint32_t nh = committed * 2u;
    if (nh < a.min_horizon) nh = a.min_horizon;
    if (nh > c->max_horizon) nh = c->max_horizon;
    const uint32_t left = c->n_ops - limit;
    if (nh > left) nh = left;
    if (committed == 0) {  // the op at base needs the exclusive executor
      c->excl = 1;
      nh = 0;
    }
    c->base[par ^ 1u] = limit;
    c->horizon[par ^ 1u] = nh;
    c->rounds += 1ull;
    c->committed += (unsigned long long)committed;
    c->planned += (unsigned long long)hor;
  }
  const uint32_t wid = wv::uni(wv::block_idx() * (wv::block_dim() >> 6) + (uint32_t)wv::wave_in_block());
  if (wid >= hor) return;
  const uint32_t idx = base + wid;
  if (idx >= limit) return;
  const dev::PlanHead h = dev::load_plan_head(&a.plans[wid]);
  dev::apply_op(a.v, h.op, h, lds[wv::wave_in_block()], &a.stats[wv::block_idx() & (kStatShards - 1)]);
}

// ---- exclusive executor: one wave runs one update alone ---------------------------------------------------
// Handles what the prefix rounds refuse (K_EXCL): the reference's global-write path
// (PCSR.cpp:1433-1437 -> insert(..., nullptr) climbing on POST-insert densities, :578-590), root
// overflow/underflow (double_list / half_list), windows > kBigWindow and long slides.  Whole-array work
// is handed back to the host as an ExclOut request.
constexpr uint32_t XF_FORCE_NOINFO = 1u;  // insert(..., nullptr): climb on post-insert densities
constexpr uint32_t XF_SKIP_COUNT = 2u;    // num_neighbors already adjusted by a previous attempt
constexpr uint32_t XF_ADD_NODE = 4u;      // op.src = new vertex id, op.dst = slot to insert the sentinel at, op.op = sentinel value
constexpr uint32_t XF_RESEARCH = 8u;      // add_node retry after double_list: search again (PCSR.cpp:539)


// Validation of an exclusive update inside a speculative epoch (me1 = stream index + 1, 0 = none): it runs when it is the
// lowest pending update, so the only thing that can make the epoch non-serialisable is a LATER update that was committed
// earlier on something this one reads or writes.  Checked exactly as for every other update — the stamps of the leaves it
// writes (padded by one leaf on both sides: an update that located its range by a sentinel this one moves has read the
// slot next to it), of the leaves its search and climb read, and of the sentinels it locates its range by.
struct XValid {
  const uint32_t *wstamp, *rstamp, *vws;
  uint32_t me1;
};
// does a[lo..hi] hold a value above thr?  (this lane's share; the caller ballots.)  An exclusive update's window can be the
// whole array — 2^19 stamps per array: one 4-byte load per trip made the executor's validation a 4 ms walk (8 ns per leaf,
// pure latency); long ranges go 16 stamps per lane per trip, four 16-byte loads in flight.
PMA_DEV bool xv_any_above(const uint32_t *a, uint64_t lo, uint64_t hi, uint32_t thr) {
  const uint64_t lane = (uint64_t)wv::lane();
  bool bad = false;
  if (hi >= lo && hi - lo >= 2048u) {
    const uint64_t al = (lo + 3u) & ~3ull;
    for (uint64_t i = lo + lane; i < al; i += 64) bad |= a[i] > thr;
    const uint4 *p4 = reinterpret_cast<const uint4 *>(a + al);
    const uint64_t n4 = (hi + 1u - al) >> 2;
    uint64_t i = lane;
    for (; i + 192u < n4; i += 256u) {
      const uint4 x = p4[i], y = p4[i + 64u], z = p4[i + 128u], w = p4[i + 192u];
      bad |= x.x > thr || x.y > thr || x.z > thr || x.w > thr || y.x > thr || y.y > thr || y.z > thr || y.w > thr;
      bad |= z.x > thr || z.y > thr || z.z > thr || z.w > thr || w.x > thr || w.y > thr || w.z > thr || w.w > thr;
    }
    for (; i < n4; i += 64u) {
      const uint4 x = p4[i];
      bad |= x.x > thr || x.y > thr || x.z > thr || x.w > thr;
    }
    lo = al + (n4 << 2);
  }
  for (uint64_t i = lo + lane; i <= hi; i += 64) bad |= a[i] > thr;
  return bad;
}
PMA_DEV bool xv_bad_writes(const View &v, const XValid &xv, uint64_t leaf_lo, uint64_t leaf_hi) {
  if (!xv.me1) return false;
  const uint64_t nleaves = v.g.N >> v.g.sh;
  if (leaf_lo > 0) leaf_lo--;
  if (leaf_hi + 1 < nleaves) leaf_hi++;
  const bool bad = xv_any_above(xv.wstamp, leaf_lo, leaf_hi, xv.me1) || xv_any_above(xv.rstamp, leaf_lo, leaf_hi, xv.me1);
  return wv::ballot(bad) != 0;
}
PMA_DEV bool xv_bad_reads(const View &v, const XValid &xv, const dev::RangeRec &rr, uint32_t src) {
  if (!xv.me1) return false;
  bool bad = false;
  for (uint32_t r = 0; r < rr.nr; r++) {  // (the ranges are in registers, lane r holding range r)
    const PlanRange pr = dev::range_at(rr, r);
    bad |= xv_any_above(xv.wstamp, pr.lo, pr.hi, xv.me1);
  }
  if (wv::lane() == 0 && (rr.sdep & 1u) && xv.vws[src] > xv.me1) bad = true;
  if (wv::lane() == 1 && (rr.sdep & 2u) && src + 1u < v.g.n && xv.vws[src + 1u] > xv.me1) bad = true;
  return wv::ballot(bad) != 0;
}

// in_wave_max: largest window the executor's own wave rebalances; larger ones go back to the host (multi-workgroup kernels)
// ops / op_index: op_index != kMax takes the update from the device-resident stream instead of `op`
PMA_KERNEL void k_exclusive(View v, Op op, const Op *ops, uint32_t op_index, uint32_t flags, ExclOut *out, StatShard *st, uint32_t in_wave_max, XValid xv) {
  PMA_SHARED uint32_t lds[3 * kLdsWindow];
  const int lane = wv::lane();
  const Geometry &g = v.g;
  const int sh = g.sh;
  const uint32_t logN = (uint32_t)g.logN;
  uint32_t result = X_DONE, rws = 0, rwl = 0, found = 0;
  if (op_index != kMax) op = ops[op_index];
  dev::RangeRec rr;
  rr.on = xv.me1 != 0;
#define PMA_X_VIOLATION()            \
  do {                               \
    if (lane == 0) {                 \
      out->result = X_VIOLATION;     \
      out->wstart = 0;               \
      out->wlen = 0;                 \
      out->found = 0;                \
    }                                \
    return;                          \
  } while (0)
  const bool add_node = (flags & XF_ADD_NODE) != 0;
  if (!add_node && op.src >= g.n) {  // silently ignored (PCSR.cpp:1375); the round planner classifies it K_NOOP, so it only
                                     // gets here through a caller's mistake — never index nodes[] with it
    if (lane == 0) {
      wv::atomic_add_u64(&st->noops, 1ull);
      out->result = X_DONE;
      out->wstart = 0;
      out->wlen = 0;
      out->found = 0;
    }
    return;
  }
  if (op.op != 0 || add_node) {
    Edge elem{op.src, op.dst, op.op};
    uint32_t index;
    if (add_node) {
      elem.dest = kMax;
      if (flags & XF_RESEARCH) {
        const Node nd = v.nodes[op.src];
        dev::SearchHit hit_;
        index = dev::pma_search(v, kMax, nd.beginning + 1, nd.end, rr, &hit_);
      } else {
        index = op.dst;
      }
    } else {
      const Node nd = v.nodes[op.src];
      dev::SearchHit hit_;
      index = dev::pma_search(v, op.dst, nd.beginning + 1, nd.end, rr, &hit_);
      if (!(flags & XF_SKIP_COUNT) && lane == 0) {
        v.vdirty[op.src] = v.serial;
        wv::atomic_add_nn(&v.nodes[op.src].num_neighbors, 1u);
      }
    }
    const Edge at = v.items[index];
    wv::fence();  // every lane has read the slot before lane 0 may overwrite it
    const bool occupied = !is_null(at);
    if (occupied && !is_sentinel(elem) && at.dest == elem.dest) {  // PCSR.cpp:529-532
      if (xv_bad_reads(v, xv, rr, op.src) || xv_bad_writes(v, xv, index >> sh, index >> sh)) PMA_X_VIOLATION();
      if (lane == 0) {
        v.items[index].value = elem.value;
        v.ldirty[index >> sh] = v.serial;
        wv::atomic_add_u64(&st->duplicates, 1ull);
      }
    } else if (occupied && (uint64_t)index == g.N - 1) {  // PCSR.cpp:533-540
      result = X_DOUBLE_THEN_RETRY;
    } else {
      int status = dev::PS_GLOBAL_NOINFO;
      dev::InsertPlan ip;
      ip.gap = index;
      ip.max_len = logN;
      ip.node_index_final = 0;
      if (!(flags & XF_FORCE_NOINFO) && !add_node) {
        ip = dev::plan_insert<true>(v, index, occupied, v.leafcnt[index >> g.sh],
                              occupied ? dev::find_gap_right(v, index + 1, kMaxSlide) : index, rr);
        status = ip.status;
      }
      if (status == dev::PS_SLIDE_OFF_END || status == dev::PS_SLIDE_LONG) status = dev::PS_OK;  // (the window plan is complete)
      uint32_t gap = index;
      bool off_end = false;
      if (occupied) {
        gap = dev::find_gap_right(v, index + 1, kMax - 1u);
        off_end = ((uint64_t)gap == g.N);
      }
      uint32_t gleft = kMax;
      if (off_end) gleft = (index >= 2) ? dev::find_gap_left(v, index - 2u) : kMax;
      if (off_end && gleft == kMax) {
        result = X_UNSUPPORTED;  // no null slot on either side: the reference doubles and slides from slot 0 (PCSR.cpp:378-383)
      } else {
        {  // validation: search / climb reads, the slide range and the window known so far
          uint64_t lo = index, hi = off_end ? g.N - 1 : gap;
          if (off_end && gleft < lo) lo = gleft;
          uint64_t pws = ((uint64_t)index >> sh) << sh, pwn = logN;
          if (status == dev::PS_OK && ip.max_len > logN) {
            pws = ip.node_index_final;
            pwn = ip.max_len;
          }
          if (pws < lo) lo = pws;
          if (pws + pwn - 1 > hi) hi = pws + pwn - 1;
          const uint64_t two = ((uint64_t)index) & ~(2ull * logN - 1);  // (the 2-leaf pass of a leaf that becomes full)
          if (two < lo) lo = two;
          if (two + 2ull * logN - 1 > hi) hi = two + 2ull * logN - 1;
          if (hi > g.N - 1) hi = g.N - 1;
          if (xv_bad_reads(v, xv, rr, op.src) || xv_bad_writes(v, xv, lo >> sh, hi >> sh)) PMA_X_VIOLATION();
          dev::mark_leaves(v, lo >> sh, hi >> sh);  // (dirty tags: the slide range and every window known so far)
        }
        wv::fence();  // planning reads are complete in every lane before the state is modified
        if (off_end) {
          // The slide ran off the end of the array (PCSR.cpp:347-351).  The reference slides everything back (its
          // slide_left from slot N-1 restores [index, N-1] exactly), then insert() steps one slot to the left and slides
          // THAT way (PCSR.cpp:541-544): the block [gleft+1, index-1] moves one slot left and the element lands on
          // index-1.  One side effect survives: the element of slot N-1 went through fix_sentinel(.., N) on the way out
          // and is written back without one, so a sentinel there keeps the out-of-range position N in nodes[].
          const Edge last = v.items[g.N - 1];
          dev::slide_left_wave(v, gleft, index - 1u);
          if (lane == 0) {
            v.items[index - 1u] = elem;
            dev::fix_sentinel(v, last, (uint32_t)g.N);
            wv::atomic_add_u64(&st->slide_slots, (unsigned long long)(g.N - 1 - index));
          }
          wv::fence();
          for (uint64_t lf = (uint64_t)(gleft >> sh) + (uint64_t)lane; lf <= (uint64_t)((index - 1u) >> sh); lf += 64) {
            uint32_t cnt = 0;  // recount the leaves the left slide touched
            for (uint32_t q = 0; q < logN; q++) cnt += (v.items[(lf << sh) + q].value != 0) ? 1u : 0u;
            v.leafcnt[lf] = cnt;
          }
          wv::fence();
        } else {
          if (gap != index) dev::slide_right_wave(v, index, gap);
          if (lane == 0) {
            v.items[index] = elem;
            v.leafcnt[gap >> sh] += 1u;
            wv::atomic_add_u64(&st->slide_slots, (unsigned long long)(gap - index));
          }
          wv::fence();
        }
        const uint32_t leaf = index >> sh;
        const uint32_t cpost = v.leafcnt[leaf];
        uint64_t ws, wn;
        if (cpost == logN) {
          wn = 2ull * logN;
          ws = ((uint64_t)index) & ~(wn - 1);
        } else {
          wn = logN;
          ws = (uint64_t)leaf << sh;
        }
        unsigned long long acalls = 1, aslots = wn;
        if (status == dev::PS_GLOBAL_DOUBLE) {
          result = X_NEED_DOUBLE;
        } else if (status == dev::PS_OK) {
          if (ip.max_len > logN) {
            ws = ip.node_index_final;
            wn = ip.max_len;
            acalls = 2;
            aslots += wn;
          }
        } else {  // PS_GLOBAL_NOINFO: climb on post-insert densities (PCSR.cpp:578-590)
          // The first density the reference looks at is that of (node_index, logN) AFTER the leaf / 2-leaf
          // pass (PCSR.cpp:555-564): with a 2-leaf pass that is the evened-out left leaf, so the inner pass
          // must really be executed before climbing (it cannot be folded into the outer pass here).
          dev::redistribute_wave(v, ws, wn, lds);
          uint64_t node_index = ws, len = logN;
          int level = g.H;
          uint32_t c = v.leafcnt[node_index >> sh];
          while ((uint64_t)c >= (uint64_t)g.t_up[level]) {
            len *= 2;
            if (len <= g.N) {
              level--;
              const uint64_t new_idx = node_index & ~(len - 1);
              if (new_idx < node_index) {
                c += dev::count_window_t<true>(v, new_idx, len / 2);
                node_index = new_idx;
              } else {
                c += dev::count_window_t<true>(v, new_idx + len / 2, len / 2);
              }
            } else {
              result = X_NEED_DOUBLE;
              break;
            }
          }
          if (result == X_DONE && len > logN) {
            ws = node_index;
            wn = len;
            acalls = 2;
            aslots += wn;
            if (xv_bad_writes(v, xv, ws >> sh, (ws + wn - 1) >> sh)) PMA_X_VIOLATION();  // (the rollback restores what was done so far)
            dev::mark_leaves(v, ws >> sh, (ws + wn - 1) >> sh);
          }
        }
        if (lane == 0) {
          wv::atomic_add_u64(&st->redistribute_calls, acalls);
          wv::atomic_add_u64(&st->redistribute_slots, aslots);
        }
        if (result == X_DONE) {
          if (ws + wn > g.N) {
            result = X_WINDOW_BEYOND_ARRAY;
          } else if (wn <= in_wave_max) {
            dev::redistribute_wave(v, ws, wn, lds);
          } else {
            result = X_NEED_REDIST;
            rws = (uint32_t)ws;
            rwl = (uint32_t)wn;
          }
        }
      }
    }
  } else {  // delete
    if (op.src < g.n) {
      const Node nd = v.nodes[op.src];
      dev::SearchHit hit_;
      const uint32_t index = dev::pma_search(v, op.dst, nd.beginning + 1, nd.end, rr, &hit_);
      if (!(flags & XF_SKIP_COUNT) && lane == 0) {
        v.vdirty[op.src] = v.serial;
        wv::atomic_add_nn(&v.nodes[op.src].num_neighbors, 0xFFFFFFFFu);
      }
      const Edge at = v.items[index];
      wv::fence();
      const Edge elem{op.src, op.dst, 1u};
      if (is_null(at) || is_sentinel(elem) || at.dest != op.dst) {
        if (lane == 0) wv::atomic_add_u64(&st->not_found, 1ull);
      } else {
        found = 1;
        const dev::RemovePlan rp = dev::plan_remove<true>(v, index, rr);
        if (xv_bad_reads(v, xv, rr, op.src) ||
            (!rp.half && xv_bad_writes(v, xv, rp.wstart >> sh, (rp.wstart + rp.wlen - 1) >> sh)))
          PMA_X_VIOLATION();
        wv::fence();  // planning reads are complete in every lane before the state is modified
        if (!rp.half) dev::mark_leaves(v, rp.wstart >> sh, (rp.wstart + rp.wlen - 1) >> sh);
        if (lane == 0) v.ldirty[index >> sh] = v.serial;
        if (lane == 0) {
          v.items[index].value = 0;
          v.items[index].dest = 0;
          v.leafcnt[index >> sh] -= 1u;
        }
        wv::fence();
        if (rp.half) {
          result = X_NEED_HALF;
          if (lane == 0) {
            wv::atomic_add_u64(&st->redistribute_calls, 1ull);
            wv::atomic_add_u64(&st->redistribute_slots, (unsigned long long)logN);
          }
        } else {
          if (lane == 0) {
            wv::atomic_add_u64(&st->redistribute_calls, 2ull);
            wv::atomic_add_u64(&st->redistribute_slots, (unsigned long long)logN + rp.wlen);
          }
          if (rp.wlen <= in_wave_max) {
            dev::redistribute_wave(v, rp.wstart, rp.wlen, lds);
          } else {
            result = X_NEED_REDIST;
            rws = (uint32_t)rp.wstart;
            rwl = (uint32_t)rp.wlen;
          }
        }
      }
    }
  }
  if (lane == 0) {
    out->result = result;
    out->wstart = rws;
    out->wlen = rwl;
    out->found = found;
  }
}

}  // namespace ppcsr
