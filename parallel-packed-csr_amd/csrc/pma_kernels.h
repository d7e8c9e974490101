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
      const uint32_t _lo = (pl)->rlo[_r], _hi = (pl)->rhi[_r];                              \
      if (_hi - _lo < dev::kLongRange)                                                      \
        for (uint32_t LEAFVAR = _lo; LEAFVAR <= _hi; LEAFVAR++) { BODY; }                   \
    }                                                                                       \
    if ((pl)->nlong) { /* rare: long ranges are walked by all lanes together */             \
      for (uint32_t _r = 0; _r < _nr; _r++) {                                               \
        const uint32_t _lo = (pl)->rlo[_r], _hi = (pl)->rhi[_r];                            \
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
  const uint32_t wid = wv::block_idx() * (wv::block_dim() >> 6) + (uint32_t)wv::wave_in_block();
  if (wid >= hor) return;
  const uint32_t idx = base + wid;
  const Op op = a.ops[idx];
  Plan *pl = &a.plans[wid];
  const dev::PlanRegs pr = dev::plan_op(a.v, op, pl);
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
  const uint32_t wid = wv::block_idx() * (wv::block_dim() >> 6) + (uint32_t)wv::wave_in_block();
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
  const uint32_t wid = wv::block_idx() * (wv::block_dim() >> 6) + (uint32_t)wv::wave_in_block();
  if (wid >= hor) return;
  const uint32_t idx = base + wid;
  if (idx >= limit) return;
  const Op op = a.ops[idx];
  dev::apply_op(a.v, op, &a.plans[wid], lds[wv::wave_in_block()], &a.stats[wv::block_idx() & (kStatShards - 1)]);
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
  Plan *scratch_plan;  // receives the read ranges of the search
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
  wv::fence();  // the ranges were recorded by lane 0 (rec_range): its stores before every lane's loads of them
  bool bad = false;
  const uint32_t nr = rr.nr < (uint32_t)kMaxR ? rr.nr : (uint32_t)kMaxR;
  for (uint32_t r = 0; r < nr; r++) {
    const uint32_t lo = xv.scratch_plan->rlo[r], hi = xv.scratch_plan->rhi[r];
    bad |= xv_any_above(xv.wstamp, lo, hi, xv.me1);
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
  rr.plan = xv.me1 ? xv.scratch_plan : (Plan *)nullptr;
  rr.nr = 0;
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

// ---- whole-array kernels ---------------------------------------------------------------------------------
// 12-byte null pattern {0xFFFFFFFF,0,0} written as a dword stream (coalesced)
PMA_KERNEL void k_fill_null(Edge *items, uint64_t start, uint64_t len) {
  uint32_t *w = reinterpret_cast<uint32_t *>(items + start);
  const uint64_t total = len * 3ull;
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t i = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); i < total; i += stride)
    w[i] = (i % 3ull == 0) ? kMax : 0u;
}

// leafcnt[leaf] for leaves [leaf_lo, leaf_lo+nleaves): one wave per 64 slots
PMA_KERNEL void k_recount(View v, uint64_t slot_lo, uint64_t nslots) {
  const int lane = wv::lane();
  const uint32_t logN = (uint32_t)v.g.logN;
  const int sh = v.g.sh;
  const uint64_t nchunks = (nslots + 63) / 64;
  const uint64_t wstride = (uint64_t)wv::grid_dim() * (wv::block_dim() >> 6);
  for (uint64_t ch = (uint64_t)wv::block_idx() * (wv::block_dim() >> 6) + wv::wave_in_block(); ch < nchunks; ch += wstride) {
    const uint64_t s = slot_lo + ch * 64 + (uint64_t)lane;
    bool nn = false;
    if (s < slot_lo + nslots) nn = v.items[s].value != 0;
    const uint64_t occ = wv::ballot(nn);
    const uint32_t nleaf = (logN >= 64) ? 1u : (64u >> sh);
    if ((uint32_t)lane < nleaf) {
      const uint64_t ls = slot_lo + ch * 64 + (uint64_t)lane * logN;
      if (ls < slot_lo + nslots) {
        const uint64_t sub = (logN >= 64) ? occ : ((occ >> ((uint32_t)lane * logN)) & ((1ull << logN) - 1ull));
        v.leafcnt[ls >> sh] = (uint32_t)wv::popc64(sub);
      }
    }
  }
}

// place the initial sentinels (constructor, PCSR.cpp:815-837): sentinel k sits at nodes[k].beginning
PMA_KERNEL void k_place_sentinels(View v) {
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t k = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); k < v.g.n; k += stride) {
    Edge e;
    e.src = (uint32_t)k;
    e.dest = kMax;
    e.value = (k == 0) ? kMax : (uint32_t)k;
    v.items[v.nodes[k].beginning] = e;
  }
}

// exclusive prefix sum of leafcnt over [leaf_lo, leaf_lo + nleaves) -> rank[i]; three small kernels
constexpr uint32_t kScanTile = 1024;  // leaves per workgroup
PMA_KERNEL void k_scan_tiles(const uint32_t *cnt, uint64_t nleaves, uint32_t *tilesum) {
  PMA_SHARED uint32_t red[4];
  const uint64_t b = wv::block_idx();
  uint32_t s = 0;
  for (uint32_t i = wv::thread_idx(); i < kScanTile; i += wv::block_dim()) {
    const uint64_t l = b * kScanTile + i;
    if (l < nleaves) s += cnt[l];
  }
  s = wv::reduce_add(s);
  if (wv::lane() == 0) red[wv::wave_in_block()] = s;
  wv::block_sync();
  if (wv::thread_idx() == 0) tilesum[b] = red[0] + red[1] + red[2] + red[3];
}
// ---- partial window rebalanced IN PLACE (no scratch copy, no copy-back) ----------------------------------------------
// The reference spreads a window inside the array itself (PCSR.cpp:207-247: pack to the left, then place right to left).
// Here a tile of kIpChunks x 4 x 64 source slots is held in the registers of one workgroup: the workgroup loads its tile,
// PUBLISHES that it has done so, waits until every tile whose source slots its own destination range covers has published
// too, and only then writes elements and null runs.  Both maps (k-th live element -> source slot, -> destination slot) are
// monotone, so tile i's destination range [c_i, d_i) is contiguous, d_i = c_{i+1}, and at every tile boundary the flow
// goes one way: "R" (d_i beyond tile i's last source slot: tile i writes over sources of tiles i+1...) or "L" (tile i+1
// writes over sources of tiles ...i).  A tile waits only for tiles further along its own run of R (or L) boundaries, so
//   key(i) = max(#consecutive R boundaries starting at i|i+1, #consecutive L boundaries ending at i-1|i)
// is strictly larger than the key of every tile that tile i waits for.  rb_order_body (the tail of k_scan_tilesums) sorts
// the tiles by key; k_rb_inplace workgroups draw tickets in that order — whoever a workgroup waits for drew an earlier ticket, is resident (or done) and
// publishes without waiting for anybody: no deadlock whatever the number of resident workgroups.  (A bounded spin turns a
// broken order into an error flag instead of a hang.)
constexpr uint32_t kIpMaxTiles = 8192, kIpOrderThreads = 1024;
constexpr uint32_t kIpHdrWords = 32 * 9;  // words before the order list in the engine's buffer
constexpr uint32_t kIpTicketStride = 32;  // ctl[1]: sticky error flag; ctl[kIpTicketStride * (1 + x)]: ticket counter of XCD x
PMA_DEV void rb_order_body(const uint32_t *tile_excl, const ChainTable *tb, uint32_t ntiles, uint32_t tile_slots, uint32_t *order, uint32_t *ctl) {
  PMA_SHARED ChainTable stb;
  PMA_SHARED unsigned long long nr[kIpMaxTiles / 64], nl[kIpMaxTiles / 64];  // bit b: boundary b|b+1 is NOT "R" / NOT "L"
  PMA_SHARED uint32_t hist[kIpMaxTiles + 1];
  PMA_SHARED uint32_t wtot[kIpOrderThreads / 64];
  {
    const uint32_t *g = reinterpret_cast<const uint32_t *>(tb);
    uint32_t *sp = reinterpret_cast<uint32_t *>(&stb);
    const uint32_t words = (uint32_t)((sizeof(ChainTable) - sizeof(ChainSeg) * (size_t)(kMaxSeg - tb->nseg)) / 4);
    for (uint32_t i = wv::thread_idx(); i < words; i += kIpOrderThreads) sp[i] = g[i];
  }
  for (uint32_t i = wv::thread_idx(); i <= ntiles; i += kIpOrderThreads) hist[i] = 0u;
  if (wv::thread_idx() < 8u) ctl[kIpTicketStride * (1u + wv::thread_idx())] = 0u;  // the ticket counters
  wv::block_sync();
  const int lane = wv::lane(), w = wv::wave_in_block();
  const uint64_t j = stb.j, wend = stb.index + stb.len;
  const uint32_t nwords = (ntiles + 63u) / 64u;
  int hint = -1;
  for (uint32_t wd = (uint32_t)w; wd < nwords; wd += kIpOrderThreads / 64) {
    const uint32_t b = wd * 64u + (uint32_t)lane;
    bool is_r = false, is_l = false;
    if (b + 1u < ntiles) {
      const uint64_t K = tile_excl[b + 1u];
      const uint64_t D = K < j ? chain_pos(&stb, K, &hint) : wend;      // where tile b's destination range ends
      const uint64_t B = stb.index + (uint64_t)(b + 1u) * tile_slots;   // where tile b's source slots end
      is_r = D > B;
      is_l = D < B;
    }
    const uint64_t mr = wv::ballot(!is_r), ml = wv::ballot(!is_l);
    if (lane == 0) {
      nr[wd] = mr;
      nl[wd] = ml;
    }
  }
  wv::block_sync();
  constexpr uint32_t kPer = kIpMaxTiles / kIpOrderThreads;
  uint32_t key[kPer];
#pragma unroll
  for (uint32_t r = 0; r < kPer; r++) {
    const uint32_t i = r * kIpOrderThreads + wv::thread_idx();
    key[r] = 0;
    if (i >= ntiles) continue;
    uint32_t wd = i >> 6;  // first boundary >= i that is not R (bit ntiles-1 is always set)
    uint64_t m = nr[wd] >> (i & 63u);
    uint32_t nb;
    if (m) nb = i + (uint32_t)wv::ctz64(m);
    else {
      do wd++; while (nr[wd] == 0ull);
      nb = wd * 64u + (uint32_t)wv::ctz64(nr[wd]);
    }
    const uint32_t d_r = nb - i;
    uint32_t d_l = 0;
    if (i > 0) {  // last boundary <= i-1 that is not L (none: every boundary down to tile 0 is L)
      const uint32_t b = i - 1u;
      int wl = (int)(b >> 6);
      m = nl[wl] << (63u - (b & 63u));
      if (m) d_l = (uint32_t)wv::clz64(m);
      else {
        do wl--; while (wl >= 0 && nl[wl] == 0ull);
        d_l = wl < 0 ? i : b - ((uint32_t)wl * 64u + 63u - (uint32_t)wv::clz64(nl[wl]));
      }
    }
    key[r] = d_r > d_l ? d_r : d_l;
    wv::atomic_add_u32(&hist[key[r]], 1u);  // (the keys of a wave's tiles are mostly distinct: electing leaders per key value was measured slower)
  }
  wv::block_sync();
  {  // exclusive scan of hist[0 .. ntiles]
    const uint32_t total = ntiles + 1u, per = (total + kIpOrderThreads - 1u) / kIpOrderThreads;
    const uint32_t lo = wv::thread_idx() * per, hi = lo + per < total ? lo + per : total;
    uint32_t mine = 0;
    for (uint32_t i = lo; i < hi; i++) mine += hist[i];
    uint32_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = wv::shfl(incl, lane - o < 0 ? 0 : lane - o);
      if (lane >= o) incl += y;
    }
    if (lane == 63) wtot[w] = incl;
    wv::block_sync();
    uint32_t run = incl - mine;
    for (int q = 0; q < w; q++) run += wtot[q];
    for (uint32_t i = lo; i < hi; i++) {
      const uint32_t x = hist[i];
      hist[i] = run;
      run += x;
    }
  }
  wv::block_sync();
#pragma unroll
  for (uint32_t r = 0; r < kPer; r++) {
    const uint32_t i = r * kIpOrderThreads + wv::thread_idx();
    if (i < ntiles) order[wv::atomic_add_u32(&hist[key[r]], 1u)] = i;
  }
}


constexpr uint32_t kTileSumThreads = 1024;
PMA_KERNEL void k_scan_tilesums(uint32_t *tilesum, uint64_t ntiles, unsigned long long *total, ChainTable *tb,
                                uint64_t tb_index, uint64_t tb_len, uint32_t *order, uint32_t *ctl, uint32_t tile_slots) {
  // ONE workgroup of kTileSumThreads.  Each thread owns a contiguous run of tile sums (independent loads, all in
  // flight together); waves combine through LDS; the prefix is written back while one lane builds the rebalance's exact
  // position table from the grand total (saves a launch).
  PMA_SHARED uint32_t wtot[kTileSumThreads / 64];
  const int lane = wv::lane(), w = wv::wave_in_block();
  const uint64_t per = (ntiles + kTileSumThreads - 1) / kTileSumThreads;
  const uint64_t lo = (uint64_t)wv::thread_idx() * per;
  const uint64_t hi = (lo + per < ntiles) ? lo + per : ntiles;
  uint32_t mine = 0;
  for (uint64_t i = lo; i < hi; i++) mine += tilesum[i];
  uint32_t incl = mine;
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = wv::shfl(incl, lane - o < 0 ? 0 : lane - o);
    if (lane >= o) incl += y;
  }
  if (lane == 63) wtot[w] = incl;
  wv::block_sync();  // also: every read of the un-scanned sums is done before anyone overwrites them
  uint32_t woff = 0, grand = 0;
  for (int q = 0; q < (int)(kTileSumThreads / 64); q++) {
    const uint32_t x = wtot[q];
    if (q < w) woff += x;
    grand += x;
  }
  if (wv::thread_idx() == kTileSumThreads - 1) {  // (this thread's own run is the shortest or empty)
    *total = grand;
    if (tb) build_chain_table(tb_index, tb_len, (uint64_t)grand, tb);
  }
  uint32_t run = woff + incl - mine;
  for (uint64_t i = lo; i < hi; i++) {
    const uint32_t x = tilesum[i];
    tilesum[i] = run;
    run += x;
  }
  if (order != nullptr) {  // in-place window: the order in which its tiles may be taken (needs the scanned sums and the table)
    wv::block_sync();
    rb_order_body(tilesum, tb, (uint32_t)ntiles, tile_slots, order, ctl);
  }
}
PMA_KERNEL void k_scan_apply(const uint32_t *cnt, uint64_t nleaves, const uint32_t *tilesum, uint32_t *rank) {
  PMA_SHARED uint32_t wsum[4];
  const uint64_t b = wv::block_idx();
  const int lane = wv::lane(), w = wv::wave_in_block();
  uint32_t run = tilesum[b];
  for (uint32_t it = 0; it < kScanTile / 256; it++) {
    const uint64_t l = b * kScanTile + it * 256 + wv::thread_idx();
    const uint32_t x = (l < nleaves) ? cnt[l] : 0u;
    uint32_t incl = x;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = wv::shfl(incl, lane - o < 0 ? 0 : lane - o);
      if (lane >= o) incl += y;
    }
    if (lane == 63) wsum[w] = incl;
    wv::block_sync();
    uint32_t woff = 0;
    for (int q = 0; q < w; q++) woff += wsum[q];
    if (l < nleaves) rank[l] = run + woff + incl - x;
    run += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    wv::block_sync();
  }
}

// Fused rebalance scatter: every live element of src window [src_lo, src_lo+src_len) goes to dst[pos_k - dst_bias]
// where k = its rank among the live elements (rank[] = exclusive leaf prefix) and pos_k comes from the exact chain
// table; `v` carries the NEW geometry (n, N) for the sentinel back-pointers.  Every wave also writes the null slots that follow its elements
// (element k owns output slots [pos_k, pos_{k+1})), so the destination needs no separate fill pass, every output slot
// is written exactly once, and the wave's output stretch is staged in LDS and stored as one coalesced run.
// Leaf counts of the destination are accumulated with one atomicAdd per element (dst leafcnt must be zeroed first).
constexpr uint32_t kStageSlots = 384;  // LDS staging tile per wave (4.5 KB): 64 elements at step <= 6
PMA_KERNEL void k_scatter_fill(View v, const Edge *src, uint64_t src_lo, uint64_t src_len, int src_sh, const uint32_t *rank,
                               const ChainTable *tb, Edge *dst, uint64_t dst_bias, uint32_t *dst_leafcnt, int dst_sh,
                               uint64_t dst_leaf_bias) {
  PMA_SHARED ChainTable stb;
  PMA_SHARED uint32_t stage[4][3 * kStageSlots];
  {
    const uint32_t *g = reinterpret_cast<const uint32_t *>(tb);
    uint32_t *s = reinterpret_cast<uint32_t *>(&stb);
    for (uint32_t i = wv::thread_idx(); i < sizeof(ChainTable) / 4; i += wv::block_dim()) s[i] = g[i];
  }
  wv::block_sync();
  const int lane = wv::lane();
  uint32_t *ls = stage[wv::wave_in_block()], *ld = ls + kStageSlots, *lv = ls + 2 * kStageSlots;
  const uint32_t slogN = 1u << src_sh;
  const uint64_t j = stb.j;
  const uint64_t wend = stb.index + stb.len;  // end of the destination window (absolute slot)
  const uint64_t nchunks = (src_len + 63) / 64;
  const uint64_t wstride = (uint64_t)wv::grid_dim() * (wv::block_dim() >> 6);
  if (j == 0) {  // empty window: nothing owns the output slots, null them all
    const uint64_t tstride = (uint64_t)wv::grid_dim() * wv::block_dim();
    for (uint64_t t = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); t < stb.len; t += tstride)
      dst[stb.index + t - dst_bias] = null_edge();
    return;
  }
  int hint = -1, hint2 = -1;
  for (uint64_t ch = (uint64_t)wv::block_idx() * (wv::block_dim() >> 6) + wv::wave_in_block(); ch < nchunks; ch += wstride) {
    const uint64_t off = ch * 64 + (uint64_t)lane;
    Edge e = null_edge();
    if (off < src_len) e = src[src_lo + off];
    const bool nn = e.value != 0;
    const uint64_t m = wv::ballot(nn);
    if (m == 0) continue;
    uint64_t pos = 0, nxt = 0;
    if (nn) {
      const uint64_t lleaf = off >> src_sh;
      uint64_t lmask;
      if (slogN >= 64) {
        lmask = ~0ull;
      } else {
        const uint32_t first = (uint32_t)(lane & ~(int)(slogN - 1));
        lmask = ((1ull << slogN) - 1ull) << first;
      }
      const uint64_t k = (uint64_t)rank[lleaf] + (uint64_t)wv::popc64(m & lmask & ((1ull << lane) - 1ull));
      pos = chain_pos(&stb, k, &hint);
      nxt = (k + 1 < j) ? chain_pos(&stb, k + 1, &hint2) : wend;
      dev::fix_sentinel(v, e, (uint32_t)pos);
    }
    // output stretch of this chunk: [first element's pos, last element's nxt)
    const int lfirst = wv::ctz64(m), llast = 63 - __builtin_clzll(m);
    const uint64_t o_lo = ((uint64_t)wv::shfl((uint32_t)(pos >> 32), lfirst) << 32) | wv::shfl((uint32_t)pos, lfirst);
    const uint64_t o_hi = ((uint64_t)wv::shfl((uint32_t)(nxt >> 32), llast) << 32) | wv::shfl((uint32_t)nxt, llast);
    const uint64_t p_hi = ((uint64_t)wv::shfl((uint32_t)(pos >> 32), llast) << 32) | wv::shfl((uint32_t)pos, llast);
    // destination leaf counts: the chunk's elements land in a handful of consecutive leaves -> one atomic per leaf
    {
      const uint64_t l0 = o_lo >> dst_sh, l1 = p_hi >> dst_sh;
      const uint64_t mylf = pos >> dst_sh;
      for (uint64_t L = l0; L <= l1; L++) {
        const uint64_t mm = wv::ballot(nn && mylf == L);
        if (mm && lane == 0) wv::atomic_add_u32(&dst_leafcnt[L - dst_leaf_bias], (uint32_t)wv::popc64(mm));
      }
    }
    const uint64_t olen = o_hi - o_lo;
    if (olen <= kStageSlots) {
      for (uint32_t t = (uint32_t)lane; t < (uint32_t)olen; t += 64) {
        ls[t] = kMax;
        ld[t] = 0;
        lv[t] = 0;
      }
      wv::lds_fence();
      if (nn) {
        const uint32_t t = (uint32_t)(pos - o_lo);
        ls[t] = e.src;
        ld[t] = e.dest;
        lv[t] = e.value;
      }
      wv::lds_fence();
      for (uint32_t t = (uint32_t)lane; t < (uint32_t)olen; t += 64) {
        Edge o;
        o.src = ls[t];
        o.dest = ld[t];
        o.value = lv[t];
        dst[o_lo + t - dst_bias] = o;
      }
      wv::lds_fence();
    } else if (nn) {  // very sparse destination: each lane writes its own run
      dst[pos - dst_bias] = e;
      for (uint64_t s2 = pos + 1; s2 < nxt; s2++) dst[s2 - dst_bias] = null_edge();
    }
  }
}

// Leaner variant of the fused rebalance pass: ONE position-table look-up per wave (its <= 64 live elements are
// consecutive ranks and almost always lie on one arithmetic progression of the table: pos_i = (A + i*D) >> shift),
// and every lane stores its own run — the element followed by the null slots up to the next element's position —
// straight from registers.  Consecutive lanes write consecutive runs, so a wave's stores cover one contiguous stretch.
PMA_KERNEL void k_scatter_runs(View v, const Edge *src, uint64_t src_lo, uint64_t src_len, int src_sh, const uint32_t *rank,
                               const ChainTable *tb, Edge *dst, uint64_t dst_bias, uint32_t *dst_leafcnt, int dst_sh,
                               uint64_t dst_leaf_bias) {
  PMA_SHARED ChainTable stb;
  {  // only the segments in use are copied (a table has <= ~30 of its 128 slots filled)
    const uint32_t *g = reinterpret_cast<const uint32_t *>(tb);
    uint32_t *s = reinterpret_cast<uint32_t *>(&stb);
    const uint32_t words = (uint32_t)((sizeof(ChainTable) - sizeof(ChainSeg) * (size_t)(kMaxSeg - tb->nseg)) / 4);
    for (uint32_t i = wv::thread_idx(); i < words; i += wv::block_dim()) s[i] = g[i];
  }
  wv::block_sync();
  const int lane = wv::lane();
  const uint64_t j = stb.j;
  const uint64_t wend = stb.index + stb.len;
  const uint64_t nchunks = (src_len + 63) / 64;
  const uint64_t wstride = (uint64_t)wv::grid_dim() * (wv::block_dim() >> 6);
  if (j == 0) {
    const uint64_t tstride = (uint64_t)wv::grid_dim() * wv::block_dim();
    for (uint64_t t = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); t < stb.len; t += tstride)
      dst[stb.index + t - dst_bias] = null_edge();
    return;
  }
  int hint = -1, hint2 = -1, hint3 = -1;
  for (uint64_t ch = (uint64_t)wv::block_idx() * (wv::block_dim() >> 6) + wv::wave_in_block(); ch < nchunks; ch += wstride) {
    const uint64_t off = ch * 64 + (uint64_t)lane;
    Edge e = null_edge();
    if (off < src_len) e = src[src_lo + off];
    const uint64_t k0 = rank[(ch * 64) >> src_sh];  // live elements before this (leaf-aligned) chunk
    const bool nn = e.value != 0;
    const uint64_t m = wv::ballot(nn);
    if (m == 0) continue;
    const uint32_t cnt = (uint32_t)wv::popc64(m);
    const uint32_t i = dev::lanemask_lt_count(m, lane);
    uint64_t A, D;
    int shift;
    uint64_t pos = 0, nxt = 0;
    if (chain_linear_run(&stb, k0, (k0 + cnt <= j - 1) ? cnt : cnt - 1, &hint3, &A, &D, &shift)) {
      pos = (A + (uint64_t)i * D) >> shift;
      nxt = (k0 + i + 1 < j) ? ((A + (uint64_t)(i + 1) * D) >> shift) : wend;
    } else if (nn) {
      pos = chain_pos(&stb, k0 + i, &hint);
      nxt = (k0 + i + 1 < j) ? chain_pos(&stb, k0 + i + 1, &hint2) : wend;
    }
    if (nn) {
      dst[pos - dst_bias] = e;
      for (uint64_t s2 = pos + 1; s2 < nxt; s2++) dst[s2 - dst_bias] = null_edge();
      dev::fix_sentinel(v, e, (uint32_t)pos);
    }
    // destination leaf counts: one atomic per leaf touched by this wave
    const int lfirst = wv::ctz64(m), llast = 63 - __builtin_clzll(m);
    const uint64_t l0 = (((uint64_t)wv::shfl((uint32_t)(pos >> 32), lfirst) << 32) | wv::shfl((uint32_t)pos, lfirst)) >> dst_sh;
    const uint64_t l1 = (((uint64_t)wv::shfl((uint32_t)(pos >> 32), llast) << 32) | wv::shfl((uint32_t)pos, llast)) >> dst_sh;
    const uint64_t mylf = pos >> dst_sh;
    for (uint64_t L = l0; L <= l1; L++) {
      const uint64_t mm = wv::ballot(nn && mylf == L);
      if (mm && lane == 0) wv::atomic_add_u32(&dst_leafcnt[L - dst_leaf_bias], (uint32_t)wv::popc64(mm));
    }
  }
}

// ---- rebalance with the leaf-rank scan folded in --------------------------------------------------------------------
// Tile = kRbTile source leaves per workgroup.  k_rb_tilesums: per-tile live counts (+ zeroing of the destination leaf
// counts as a side job); k_scan_tilesums: exclusive scan of the tile sums + the exact position table; k_rb_scatter: each
// workgroup scans its own tile's leaf counts in LDS (so no per-leaf rank array is ever written or read) and runs the
// register-run scatter of k_scatter_runs over the tile's chunks.  Three launches for a whole-array rebalance.
constexpr uint32_t kRbTile = 256;  // maximum tile (= workgroup size); the engine picks a power of two <= this per window
PMA_KERNEL void k_rb_tilesums(uint32_t *cnt, uint64_t nleaves, uint32_t tile_leaves, uint32_t *tilesum, uint32_t *copy_out,
                              uint32_t *zero_ptr, uint64_t zero_n, uint32_t *dirty, uint32_t serial) {
  PMA_SHARED uint32_t red[4];
  const uint64_t b = wv::block_idx();
  const uint64_t l = b * tile_leaves + wv::thread_idx();
  const bool mine = wv::thread_idx() < tile_leaves && l < nleaves;
  uint32_t s = mine ? cnt[l] : 0u;
  if (dirty != nullptr && mine) dirty[l] = serial;  // (a window of the live array is about to be rewritten: dirty tags)
  if (copy_out != nullptr && mine) {  // in-place window: park the source counts, clear them for the rebuild
    copy_out[l] = s;
    cnt[l] = 0u;
  }
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t i = b * wv::block_dim() + wv::thread_idx(); i < zero_n; i += stride) zero_ptr[i] = 0u;
  s = wv::reduce_add(s);
  if (wv::lane() == 0) red[wv::wave_in_block()] = s;
  wv::block_sync();
  if (wv::thread_idx() == 0) tilesum[b] = red[0] + red[1] + red[2] + red[3];
}
// One chunk (64 source slots, already in registers) of the rebalance scatter: rank the live slots, look their exact
// positions up, store element + trailing nulls, fix sentinels, add the destination leaf counts.
PMA_DEV void rb_scatter_chunk(const View &v, const Edge &e, uint64_t k0, const ChainTable *stb, uint64_t j, uint64_t wend,
                              Edge *__restrict__ dst, uint64_t dst_bias, uint32_t *dst_leafcnt, int dst_sh, uint64_t dst_leaf_bias,
                              int lane, uint64_t lt_mask, int *hint, int *hint2, int *hint3) {
  const bool nn = e.value != 0;
  const uint64_t m = wv::ballot(nn);
  if (m == 0) return;
  const uint32_t cn = (uint32_t)wv::popc64(m);
  const uint64_t below = m & lt_mask;
  const uint32_t i = (uint32_t)wv::popc64(below);
  uint64_t A, D;
  int shift;
  uint64_t pos = 0, nxt = 0;
  if (chain_linear_run(stb, k0, (k0 + cn <= j - 1) ? cn : cn - 1, hint3, &A, &D, &shift)) {
    const uint64_t M = A + (uint64_t)i * D;
    pos = M >> shift;
    nxt = (k0 + i + 1 < j) ? ((M + D) >> shift) : wend;
  } else if (nn) {
    pos = chain_pos(stb, k0 + i, hint);
    nxt = (k0 + i + 1 < j) ? chain_pos(stb, k0 + i + 1, hint2) : wend;
  }
  if (nn) {
    dst[pos - dst_bias] = e;
    for (uint64_t s2 = pos + 1; s2 < nxt; s2++) dst[s2 - dst_bias] = null_edge();
    dev::fix_sentinel(v, e, (uint32_t)pos);
  }
  // destination leaf counts: the first live lane of every destination leaf adds that leaf's share of this chunk
  const uint32_t mylf = (uint32_t)(pos >> dst_sh);
  const int prevlane = below ? 63 - __builtin_clzll(below) : lane;
  const uint32_t prevlf = wv::shfl(mylf, prevlane);
  const bool head = nn && (below == 0 || prevlf != mylf);
  const uint64_t hm = wv::ballot(head);
  if (head) {
    const uint64_t later_heads = hm & ~lt_mask & ~(1ull << lane);
    const uint64_t upto = later_heads ? ((1ull << wv::ctz64(later_heads)) - 1ull) : ~0ull;
    wv::atomic_add_u32(&dst_leafcnt[(uint64_t)mylf - dst_leaf_bias], (uint32_t)wv::popc64(m & upto & ~lt_mask));
  }
}

// kRbBatch chunks are requested back to back before the first one is processed: the kernel is bound by memory latency
// per wave (load -> rank -> store -> store acknowledgement), so bytes in flight per wave are what buys bandwidth.
constexpr int kRbBatch = 4;
PMA_KERNEL void k_rb_scatter(View v, const Edge *__restrict__ src, uint64_t src_lo, uint64_t src_len, int src_sh,
                             const uint32_t *__restrict__ cnt, uint32_t tile_leaves, uint32_t batch,
                             const uint32_t *__restrict__ tile_excl, const ChainTable *tb, Edge *__restrict__ dst, uint64_t dst_bias,
                             uint32_t *dst_leafcnt, int dst_sh, uint64_t dst_leaf_bias) {
  PMA_SHARED ChainTable stb;
  PMA_SHARED uint32_t pre[kRbTile];
  PMA_SHARED uint32_t wsum[4];
  {
    const uint32_t *g = reinterpret_cast<const uint32_t *>(tb);
    uint32_t *sp = reinterpret_cast<uint32_t *>(&stb);
    const uint32_t words = (uint32_t)((sizeof(ChainTable) - sizeof(ChainSeg) * (size_t)(kMaxSeg - tb->nseg)) / 4);
    for (uint32_t i = wv::thread_idx(); i < words; i += wv::block_dim()) sp[i] = g[i];
  }
  const int lane = wv::lane(), w = wv::wave_in_block();
  const uint64_t tile = wv::block_idx();
  const uint64_t nleaves = src_len >> src_sh;
  {  // exclusive prefix of this tile's leaf counts (one leaf per thread)
    const uint64_t l = tile * tile_leaves + wv::thread_idx();
    const uint32_t x = (wv::thread_idx() < tile_leaves && l < nleaves) ? cnt[l] : 0u;
    uint32_t incl = x;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = wv::shfl(incl, lane - o < 0 ? 0 : lane - o);
      if (lane >= o) incl += y;
    }
    if (lane == 63) wsum[w] = incl;
    wv::block_sync();
    uint32_t woff = 0;
    for (int q = 0; q < w; q++) woff += wsum[q];
    pre[wv::thread_idx()] = woff + incl - x;
  }
  wv::block_sync();
  const uint64_t j = stb.j;
  const uint64_t wend = stb.index + stb.len;
  if (j == 0) {
    const uint64_t tstride = (uint64_t)wv::grid_dim() * wv::block_dim();
    for (uint64_t t = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); t < stb.len; t += tstride)
      dst[stb.index + t - dst_bias] = null_edge();
    return;
  }
  const uint64_t base_rank = tile_excl[tile];
  const uint32_t lpc = 64u >> src_sh;               // leaves per 64-slot chunk (logN <= 32)
  const uint32_t chunks = tile_leaves / lpc;        // chunks in this tile
  const uint64_t tile_slot0 = (tile * tile_leaves) << src_sh;
  const uint64_t lt_mask = (1ull << lane) - 1ull;   // lanes below this one
  int hint = -1, hint2 = -1, hint3 = -1;
  if (batch >= (uint32_t)kRbBatch) {
    for (uint32_t c0 = (uint32_t)w * kRbBatch; c0 < chunks; c0 += 4 * kRbBatch) {
      Edge e[kRbBatch];
#pragma unroll
      for (int q = 0; q < kRbBatch; q++) {
        const uint64_t off = tile_slot0 + (uint64_t)(c0 + q) * 64 + (uint64_t)lane;
        e[q] = null_edge();
        if (c0 + q < chunks && off < src_len) e[q] = src[src_lo + off];
      }
#pragma unroll
      for (int q = 0; q < kRbBatch; q++) {
        if (c0 + q < chunks)
          rb_scatter_chunk(v, e[q], base_rank + pre[(c0 + q) * lpc], &stb, j, wend, dst, dst_bias, dst_leafcnt, dst_sh, dst_leaf_bias,
                           lane, lt_mask, &hint, &hint2, &hint3);
      }
    }
  } else {
    for (uint32_t c = (uint32_t)w; c < chunks; c += 4) {
      const uint64_t off = tile_slot0 + (uint64_t)c * 64 + (uint64_t)lane;
      if (off - lane >= src_len) break;
      Edge e = null_edge();
      if (off < src_len) e = src[src_lo + off];
      rb_scatter_chunk(v, e, base_rank + pre[c * lpc], &stb, j, wend, dst, dst_bias, dst_leafcnt, dst_sh, dst_leaf_bias, lane, lt_mask,
                       &hint, &hint2, &hint3);
    }
  }
}

// ---- destination-centric rebalance pass (round 3) ---------------------------------------------------------------------
// k_rb_scatter walks SOURCE tiles and stores every element (and the nulls behind it) where it goes: one 12-byte store per
// lane at scattered addresses, several store instructions per chunk.  k_rb_gather turns the pass round: a workgroup owns
// kGtSlots DESTINATION slots, finds the ranks that land there (the position table is monotone: ranks [k_lo, k_hi)), streams
// the source chunks that hold those ranks (a contiguous stretch of the source), places the elements in an LDS image of
// its destination tile that starts out as all nulls, and writes the image with full 16-byte stores — every destination
// byte leaves the CU exactly once, coalesced.  Destination leaf counts come out of the image (LDS atomics), so nobody
// zeroes or atomically adds to the global leaf counts.  Same position table, same results.
PMA_DEV uint64_t gt_pos_or_end(const ChainTable *stb, uint64_t k, uint64_t j, uint64_t wend) {
  int hint = -1;
  return k < j ? chain_pos(stb, k, &hint) : wend;
}
// first rank k in [0, j] whose position is >= T (j: none).  Whole wave; every lane returns the answer.
PMA_DEV uint64_t gt_first_rank_at(const ChainTable *stb, uint64_t T, uint64_t j, uint64_t wend, int lane) {
  if (T <= stb->index) return 0;
  if (T >= wend) return j;
  // positions are about index + k * len / j: start one wave-width window around the estimate, slide until it brackets T
  uint64_t k0 = (uint64_t)((double)(T - stb->index) * ((double)j / (double)stb->len));  // (an estimate: fp64 is plenty)
  if (k0 > j) k0 = j;
  k0 = k0 > 31 ? k0 - 31 : 0;
  for (;;) {
    if (k0 + 63 > j) k0 = j > 63 ? j - 63 : 0;
    const uint64_t k = k0 + (uint64_t)lane;  // <= j
    const bool ge = gt_pos_or_end(stb, k <= j ? k : j, j, wend) >= T;
    const uint64_t m = wv::ballot(ge);
    if (m == 0) {  // all below T: the answer lies above this window (k0 + 63 < j here, or the window's last lane is j itself)
      k0 += 64;
      continue;
    }
    const int f = wv::ctz64(m);
    if (f == 0 && k0 > 0) {  // the window's first rank is already at or past T: look further down
      k0 = k0 > 63 ? k0 - 63 : 0;
      continue;
    }
    return k0 + (uint64_t)f;
  }
}
// source tile that holds rank k: the last t with tile_excl[t] <= k (tile_excl[0] = 0).  Whole wave.  One 64-wide probe
// around the spot a uniform spread suggests, then 64-ary narrowing of whatever range is left.
PMA_DEV uint64_t gt_find_tile(const uint32_t *__restrict__ tile_excl, uint64_t ntiles, uint64_t k, uint64_t j, int lane) {
  uint64_t lo = 0, hi = ntiles;  // the answer is in [lo, hi); tile_excl[lo] <= k
  {
    uint64_t g = (uint64_t)((double)k * ((double)ntiles / (double)(j ? j : 1)));
    if (g > ntiles) g = ntiles;
    g = g > 31 ? g - 31 : 0;
    if (g + 64 > ntiles) g = ntiles > 64 ? ntiles - 64 : 0;
    const uint64_t t = g + (uint64_t)lane;
    const uint64_t m = wv::ballot(t < ntiles && (uint64_t)tile_excl[t] <= k);
    if (m == 0) {
      hi = g;  // (g > 0 here: tile_excl[0] = 0 <= k)
    } else {
      const int f = 63 - __builtin_clzll(m);
      lo = g + (uint64_t)f;
      if (f < 63 || lo + 1 >= ntiles) return lo;  // the next tile was probed and lies above k, or there is none
    }
  }
  while (hi - lo > 1) {
    const uint64_t span = (hi - lo + 63) / 64;
    const uint64_t t = lo + (uint64_t)lane * span;
    const uint64_t m = wv::ballot(t < hi && (uint64_t)tile_excl[t] <= k);  // (lane 0 is always set)
    const int f = 63 - __builtin_clzll(m);
    lo += (uint64_t)f * span;
    if (lo + span < hi) hi = lo + span;
  }
  return lo;
}
template <int SEGS>
struct ChainTableLds {  // ChainTable with a shorter segment list (same layout in front): what a workgroup keeps in LDS
  uint64_t index, len, j;
  int nseg;
  int overflow;
  ChainSeg seg[SEGS];
};
// A workgroup owns a RUN of consecutive destination tiles.  The ranks that land in consecutive tiles are consecutive, so
// after one search for the start of the run the source is simply streamed: batches of 4 x kGtBatch chunks (the next batch
// is requested before the current one is placed), live counts through LDS give every element its rank, elements below
// the current tile's last rank go into the image; when the batch runs past it the image is written out, cleared, and the
// same batch continues into the next tile.  The start-up searches are paid once per run, not once per tile.
constexpr int kGtBatch = 4;  // chunks per wave and batch
// (blockDim.x is a load from the dispatch packet wherever it is used, and the wait behind that load drains every store and
//  prefetch in flight: the workgroup size is a constant here)
constexpr uint32_t kGtThreads = 256;
template <uint32_t SLOTS, int SEGS>
PMA_DEV void rb_gather_body(const View &v, const Edge *__restrict__ src, uint64_t src_lo, uint64_t src_len, int src_sh,
                            const uint32_t *__restrict__ cnt, uint32_t tile_leaves, uint64_t ntiles_src,
                            const uint32_t *__restrict__ tile_excl, const ChainTable *tb, Edge *__restrict__ dst, uint64_t dst_bias,
                            uint32_t *dst_leafcnt, int dst_sh, uint64_t dst_leaf_bias, uint32_t run_tiles) {
  PMA_SHARED ChainTableLds<SEGS> stb_s;
  PMA_SHARED uint32_t img[SLOTS * 3];
  PMA_SHARED uint32_t lcnt[SLOTS / 4];
  PMA_SHARED uint32_t pre[256 + 1];
  PMA_SHARED uint32_t wsum[4];
  PMA_SHARED uint32_t bc[2][4 * kGtBatch];  // live elements per chunk of the batch (double buffered: one sync per batch)
  PMA_SHARED unsigned long long bnd[2];     // first chunk of the stream, rank in front of it
  const ChainTable *stb = reinterpret_cast<const ChainTable *>(&stb_s);
  const uint32_t tid = wv::thread_idx();
  const int lane = wv::lane(), w = wv::wave_in_block();
  {
    const uint32_t *g = reinterpret_cast<const uint32_t *>(tb);
    uint32_t *sp = reinterpret_cast<uint32_t *>(&stb_s);
    const uint32_t words = (uint32_t)((sizeof(ChainTable) - sizeof(ChainSeg) * (size_t)(kMaxSeg - tb->nseg)) / 4);
    for (uint32_t i = tid; i < words; i += kGtThreads) sp[i] = g[i];  // (the host picked SEGS for this window)
  }
  for (uint32_t i = tid; i < SLOTS * 3; i += kGtThreads) img[i] = (i % 3u == 0u) ? kMax : 0u;  // null_edge()
  for (uint32_t i = tid; i < SLOTS / 4; i += kGtThreads) lcnt[i] = 0u;
  wv::block_sync();
  const uint64_t j = stb->j, wend = stb->index + stb->len;
  const uint64_t ntiles_dst = (stb->len + SLOTS - 1) / SLOTS;
  uint64_t t = (uint64_t)wv::block_idx() * run_tiles;
  const uint64_t t_end = (t + run_tiles < ntiles_dst) ? t + run_tiles : ntiles_dst;
  const int lsh = 6 - src_sh;  // log2(leaves per 64-slot chunk)
  const uint64_t nleaves = src_len >> src_sh;
  const uint64_t nchunks = (src_len + 63) >> 6;
  uint64_t T0 = stb->index + t * SLOTS;
  uint64_t T1 = (T0 + SLOTS < wend) ? T0 + SLOTS : wend;
  uint64_t k_lo = gt_first_rank_at(stb, T0, j, wend, lane);  // (every wave computes it: no hand-over, no barrier)
  uint64_t k_hi = gt_first_rank_at(stb, T1, j, wend, lane);
  // where the stream starts: the source chunk that holds rank k_lo, and the rank of that chunk's first live element
  {
    const uint64_t ts = (k_lo < j && ntiles_src > 1) ? gt_find_tile(tile_excl, ntiles_src, k_lo, j, lane) : 0;
    const uint64_t base = (k_lo < j && ntiles_src > 0) ? (uint64_t)tile_excl[ts] : 0ull;
    const uint64_t l = ts * (uint64_t)tile_leaves + tid;
    const uint32_t x = (k_lo < j && tid < tile_leaves && l < nleaves) ? cnt[l] : 0u;
    uint32_t incl = x;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = wv::shfl(incl, lane - o < 0 ? 0 : lane - o);
      if (lane >= o) incl += y;
    }
    if (lane == 63) wsum[w] = incl;
    if (tid == 0) {
      bnd[0] = nchunks;  // (no element at or after k_lo: nothing to stream)
      bnd[1] = j;
    }
    wv::block_sync();
    uint32_t woff = 0;
    for (int q = 0; q < w; q++) woff += wsum[q];
    const uint32_t ex = woff + incl - x;
    pre[tid] = ex;
    wv::block_sync();
    if (x != 0 && base + ex <= k_lo && k_lo < base + ex + x) {  // the leaf that holds rank k_lo: exactly one thread
      const uint32_t lf0 = (tid >> lsh) << lsh;                  // first leaf of its chunk
      bnd[0] = (ts * (uint64_t)tile_leaves + lf0) >> lsh;
      bnd[1] = base + pre[lf0];
    }
    wv::block_sync();
  }
  uint64_t cc = bnd[0];  // next chunk to request
  uint64_t rb = bnd[1];  // rank of the first live element of chunk cc
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  int hint = -1, hint3 = -1;
  // Loads are issued unconditionally (the address is clamped, the value is masked when it is used): a load under a branch
  // makes the compiler wait for it on the spot, which serialises the batch and defeats the prefetch.
  Edge e[kGtBatch], en[kGtBatch];
  bool okn[kGtBatch];
#pragma unroll
  for (int q = 0; q < kGtBatch; q++) {
    const uint64_t off = ((cc + (uint64_t)(w * kGtBatch + q)) << 6) + (uint64_t)lane;
    okn[q] = off < src_len;
    en[q] = src[src_lo + (okn[q] ? off : 0)];
  }
  uint32_t par = 0;
  for (;;) {
#pragma unroll
    for (int q = 0; q < kGtBatch; q++) {
      e[q] = en[q];
      if (!okn[q]) e[q] = null_edge();
    }
#pragma unroll
    for (int q = 0; q < kGtBatch; q++) {  // the next batch is on its way while this one is placed
      const uint64_t off = ((cc + (uint64_t)(4 * kGtBatch + w * kGtBatch + q)) << 6) + (uint64_t)lane;
      okn[q] = off < src_len;
      en[q] = src[src_lo + (okn[q] ? off : 0)];
    }
    uint64_t m[kGtBatch];
#pragma unroll
    for (int q = 0; q < kGtBatch; q++) {
      m[q] = wv::ballot(e[q].value != 0);
      if (lane == 0) bc[par][w * kGtBatch + q] = (uint32_t)wv::popc64(m[q]);
    }
    wv::block_sync();
    uint32_t before = 0, btotal = 0;  // live elements of the batch in front of this wave's chunks / in all of it
#pragma unroll
    for (int q = 0; q < 4 * kGtBatch; q++) {
      const uint32_t c = bc[par][q];
      if (q < w * kGtBatch) before += c;
      btotal += c;
    }
    par ^= 1u;
    uint64_t kq[kGtBatch];
    uint64_t pq[kGtBatch];
    {
      uint64_t r0 = rb + before;
#pragma unroll
      for (int q = 0; q < kGtBatch; q++) {
        const uint32_t cn = (uint32_t)wv::popc64(m[q]);
        const uint32_t i = (uint32_t)wv::popc64(m[q] & lt_mask);
        const bool nn = e[q].value != 0;
        kq[q] = r0 + i;
        pq[q] = 0;
        if (cn) {
          uint64_t A, D;
          int shift;
          if (chain_linear_run(stb, r0, (r0 + cn <= j - 1) ? cn : cn - 1, &hint3, &A, &D, &shift)) pq[q] = (A + (uint64_t)i * D) >> shift;
          else if (nn) pq[q] = chain_pos(stb, kq[q], &hint);
        }
        r0 += cn;
      }
    }
    const uint64_t bend = rb + btotal;  // rank behind the batch
    for (;;) {                          // the tiles this batch reaches
#pragma unroll
      for (int q = 0; q < kGtBatch; q++) {
        if (e[q].value != 0 && kq[q] >= k_lo && kq[q] < k_hi) {
          const uint32_t o = (uint32_t)(pq[q] - T0);
          img[o * 3u] = e[q].src;
          img[o * 3u + 1u] = e[q].dest;
          img[o * 3u + 2u] = e[q].value;
          dev::fix_sentinel(v, e[q], (uint32_t)pq[q]);
        }
        // destination leaf counts: the lanes that land in one leaf are neighbours, so the first of each group adds the
        // group's size (same-address LDS atomics are served one lane at a time: one add per element cost 100 us here)
        {
          const bool mine = e[q].value != 0 && kq[q] >= k_lo && kq[q] < k_hi;
          const uint64_t mm = wv::ballot(mine);
          if (mm != 0) {
            const uint32_t lf = mine ? (uint32_t)((pq[q] - T0) >> dst_sh) : 0xFFFFFFFFu;
            const uint64_t below = mm & lt_mask;
            const int prevlane = below ? 63 - __builtin_clzll(below) : lane;
            const uint32_t prevlf = wv::shfl(lf, prevlane);
            const bool head = mine && (below == 0 || prevlf != lf);
            const uint64_t hm = wv::ballot(head);
            if (head) {
              const uint64_t later = hm & ~lt_mask & ~(1ull << lane);
              const uint64_t upto = later ? ((1ull << wv::ctz64(later)) - 1ull) : ~0ull;
              wv::atomic_add_u32(&lcnt[lf], (uint32_t)wv::popc64(mm & upto & ~lt_mask));
            }
          }
        }
      }
      if (!(bend >= k_hi || bend >= j || cc >= nchunks)) break;  // the tile may get more from the next batch
      // the tile is complete: out it goes (16-byte stores when it starts on a 16-byte boundary), then a fresh image
      wv::block_sync();
      const uint32_t nsl = (uint32_t)(T1 - T0);
      uint32_t *out = reinterpret_cast<uint32_t *>(dst + (T0 - dst_bias));
      if ((reinterpret_cast<uintptr_t>(out) & 15u) == 0 && (nsl & 3u) == 0) {
        uint4 *o4 = reinterpret_cast<uint4 *>(out);
        const uint4 *i4 = reinterpret_cast<const uint4 *>(img);
        for (uint32_t i = tid; i < nsl * 3u / 4u; i += kGtThreads) o4[i] = i4[i];
      } else {
        for (uint32_t i = tid; i < nsl * 3u; i += kGtThreads) out[i] = img[i];
      }
      const uint32_t nlf = nsl >> dst_sh;
      for (uint32_t i = tid; i < nlf; i += kGtThreads) dst_leafcnt[(T0 >> dst_sh) - dst_leaf_bias + i] = lcnt[i];
      if (++t >= t_end) return;
      wv::block_sync();
      for (uint32_t i = tid; i < SLOTS * 3; i += kGtThreads) img[i] = (i % 3u == 0u) ? kMax : 0u;
      for (uint32_t i = tid; i < SLOTS / 4; i += kGtThreads) lcnt[i] = 0u;
      wv::block_sync();
      T0 = T1;
      T1 = (T0 + SLOTS < wend) ? T0 + SLOTS : wend;
      k_lo = k_hi;
      k_hi = gt_first_rank_at(stb, T1, j, wend, lane);
    }
    rb = bend;
    cc += 4 * kGtBatch;
  }
}
constexpr uint32_t kGtSlots = 1024;
constexpr int kGtSegsSmall = 4, kGtSegsBig = kMaxSeg;
// (windows that do not start at slot 0 lie in one binade: one or two segments; windows from slot 0 cross one per binade)
PMA_KERNEL void k_rb_gather(View v, const Edge *__restrict__ src, uint64_t src_lo, uint64_t src_len, int src_sh,
                            const uint32_t *__restrict__ cnt, uint32_t tile_leaves, uint64_t ntiles_src,
                            const uint32_t *__restrict__ tile_excl, const ChainTable *tb, Edge *__restrict__ dst, uint64_t dst_bias,
                            uint32_t *dst_leafcnt, int dst_sh, uint64_t dst_leaf_bias, uint32_t run_tiles) {
  rb_gather_body<kGtSlots, kGtSegsSmall>(v, src, src_lo, src_len, src_sh, cnt, tile_leaves, ntiles_src, tile_excl, tb, dst, dst_bias, dst_leafcnt, dst_sh,
                                         dst_leaf_bias, run_tiles);
}
PMA_KERNEL void k_rb_gather_from0(View v, const Edge *__restrict__ src, uint64_t src_lo, uint64_t src_len, int src_sh,
                                  const uint32_t *__restrict__ cnt, uint32_t tile_leaves, uint64_t ntiles_src,
                                  const uint32_t *__restrict__ tile_excl, const ChainTable *tb, Edge *__restrict__ dst, uint64_t dst_bias,
                                  uint32_t *dst_leafcnt, int dst_sh, uint64_t dst_leaf_bias, uint32_t run_tiles) {
  rb_gather_body<kGtSlots, kGtSegsBig>(v, src, src_lo, src_len, src_sh, cnt, tile_leaves, ntiles_src, tile_excl, tb, dst, dst_bias, dst_leafcnt, dst_sh,
                                       dst_leaf_bias, run_tiles);
}

constexpr uint32_t kIpSpinLimit = 1u << 22;
template <int CPW>  // chunks (64 slots) per wave: the tile is 4 * CPW * 64 slots
PMA_DEV void rb_inplace_body(const View &v, uint64_t wstart, uint64_t wlen, int sh, const uint32_t *__restrict__ cnt,
                             const uint32_t *__restrict__ tile_excl, const ChainTable *tb, const uint32_t *__restrict__ order, uint32_t *ctl,
                             uint32_t *flags, uint32_t epoch, uint32_t nlists) {
  PMA_SHARED ChainTable stb;
  PMA_SHARED uint32_t pre[kRbTile];
  PMA_SHARED uint32_t wsum[4];
  PMA_SHARED uint32_t s_tile;
  constexpr uint32_t kTileSlots = 4u * CPW * 64u;
  if (wv::thread_idx() == 0) {
    // Ticket: position t * L + x of the order, drawn from the counter of this workgroup's XCD x (one counter for all would
    // hand out ~one ticket per 9 ns: same-address atomics are served one after the other).  Each of the L sub-lists is
    // consumed in order, and a workgroup turns to another XCD's list only when its own is used up, so the earliest
    // unfinished tile of the order is always held by a resident workgroup or is the next ticket of an XCD with free slots.
    // L = nlists is the number of XCD ids the engine SAW workgroups run on when it was created (k_xcc_probe: 8 on an
    // MI355X in SPX mode; 1 — a single list, safe whatever the dispatcher does — if the ids were not 0..L-1 evenly).
    const uint32_t ntiles = (uint32_t)(wlen / kTileSlots), L = nlists, xcc = wv::xcc_id() % L;
    uint32_t pos = 0xFFFFFFFFu;
    for (uint32_t a = 0; a < L && pos == 0xFFFFFFFFu; a++) {
      const uint32_t x = (xcc + a) % L;
      const uint32_t have = x < ntiles ? (ntiles - x + L - 1u) / L : 0u;
      if (have == 0u) continue;
      const uint32_t t = wv::atomic_add_u32(&ctl[kIpTicketStride * (1u + x)], 1u);
      if (t < have) pos = t * L + x;
    }
    s_tile = pos == 0xFFFFFFFFu ? pos : order[pos];
  }
  wv::block_sync();
  const int lane = wv::lane(), w = wv::wave_in_block();
  const uint32_t tile = s_tile;
  if (tile == 0xFFFFFFFFu) return;  // (more workgroups than tiles: cannot happen with the engine's launch)
  const uint64_t tile_slot0 = (uint64_t)tile * kTileSlots;
  Edge e[CPW];  // requested first: everything below overlaps with these loads
#pragma unroll
  for (int q = 0; q < CPW; q++) {
    // (the window is a whole number of tiles — the engine checks — so the load needs no guard; a load under a branch makes the
    //  compiler wait for it on the spot, which turned these CPW requests into CPW / 2 round trips)
    const uint64_t off = tile_slot0 + (uint64_t)(w * CPW + q) * 64u + (uint64_t)lane;
    e[q] = v.items[wstart + off];
  }
  {
    const uint32_t *g = reinterpret_cast<const uint32_t *>(tb);
    uint32_t *sp = reinterpret_cast<uint32_t *>(&stb);
    const uint32_t words = (uint32_t)((sizeof(ChainTable) - sizeof(ChainSeg) * (size_t)(kMaxSeg - tb->nseg)) / 4);
    for (uint32_t i = wv::thread_idx(); i < words; i += 256u /* the launch's workgroup size: blockDim.x would be a load + a wait for everything in flight */) sp[i] = g[i];
  }
  const uint32_t tile_leaves = kTileSlots >> sh;
  const uint64_t nleaves = wlen >> sh;
  uint32_t tile_cnt;
  {  // exclusive prefix of this tile's (parked) leaf counts, one leaf per thread
    const uint64_t l = (uint64_t)tile * tile_leaves + wv::thread_idx();
    const uint32_t x = (wv::thread_idx() < tile_leaves && l < nleaves) ? cnt[l] : 0u;
    uint32_t incl = x;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = wv::shfl(incl, lane - o < 0 ? 0 : lane - o);
      if (lane >= o) incl += y;
    }
    if (lane == 63) wsum[w] = incl;
    wv::block_sync();
    uint32_t woff = 0;
    for (int q = 0; q < w; q++) woff += wsum[q];
    pre[wv::thread_idx()] = woff + incl - x;
    tile_cnt = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }
  const uint64_t j = stb.j, wend = stb.index + stb.len;
  wv::wait_loads();  // EVERY wave's tile loads have returned before the barrier that precedes "tile read" (a workgroup barrier
                     // does not wait for vmcnt, and flag_publish's own wait covers wave 0 only)
  wv::block_sync();
  if (wv::thread_idx() == 0) wv::flag_publish(&flags[tile], epoch);
  if (j == 0) {  // empty window: nothing is read by anybody, every tile clears its own slots
    for (uint32_t t = wv::thread_idx(); t < kTileSlots; t += 256u /* the launch's workgroup size: blockDim.x would be a load + a wait for everything in flight */)
      if (tile_slot0 + t < wlen) v.items[wstart + tile_slot0 + t] = null_edge();
    return;
  }
  if (tile_cnt == 0) return;
  const uint64_t base_rank = tile_excl[tile];
  int hint = -1, hint2 = -1, hint3 = -1;
  if (w == 0) {  // wait for the tiles whose source slots [c, d) covers
    const uint64_t c = chain_pos(&stb, base_rank, &hint);
    const uint64_t d = base_rank + tile_cnt < j ? chain_pos(&stb, base_rank + tile_cnt, &hint2) : wend;
    const uint32_t lo = (uint32_t)((c - wstart) / kTileSlots), hi = (uint32_t)((d - 1u - wstart) / kTileSlots);
    bool bad = false;
    for (uint32_t t0 = lo; t0 <= hi; t0 += 64u) {
      const uint32_t t = t0 + (uint32_t)lane;
      uint32_t spins = 0;
      while (wv::ballot(t <= hi && t != tile && wv::flag_read(&flags[t]) != epoch) != 0ull) {
        if (++spins > kIpSpinLimit) {
          bad = true;
          break;
        }
        wv::spin_pause();
      }
    }
    if (bad && lane == 0) ctl[1] = 1u;
    wv::flag_acquire();
  }
  wv::block_sync();
  const uint32_t lpc = 64u >> sh;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int q = 0; q < CPW; q++) {
    const uint32_t c = (uint32_t)(w * CPW + q);
    if (tile_slot0 + (uint64_t)c * 64u < wlen)
      rb_scatter_chunk(v, e[q], base_rank + pre[c * lpc], &stb, j, wend, v.items, 0, v.leafcnt, v.g.sh, 0, lane, lt_mask, &hint, &hint2, &hint3);
  }
}
PMA_KERNEL void k_rb_inplace8(View v, uint64_t wstart, uint64_t wlen, int sh, const uint32_t *cnt, const uint32_t *tile_excl, const ChainTable *tb,
                              const uint32_t *order, uint32_t *ctl, uint32_t *flags, uint32_t epoch, uint32_t nlists) {
  rb_inplace_body<8>(v, wstart, wlen, sh, cnt, tile_excl, tb, order, ctl, flags, epoch, nlists);
}
// which XCD ids do workgroups of this device report, and how evenly?  (one atomic per workgroup into 8 counters)
PMA_KERNEL void k_xcc_probe(uint32_t *counts) {
  if (wv::thread_idx() == 0) wv::atomic_add_u32(&counts[wv::xcc_id() & 7u], 1u);
}
PMA_KERNEL void k_rb_inplace16(View v, uint64_t wstart, uint64_t wlen, int sh, const uint32_t *cnt, const uint32_t *tile_excl, const ChainTable *tb,
                               const uint32_t *order, uint32_t *ctl, uint32_t *flags, uint32_t epoch, uint32_t nlists) {
  rb_inplace_body<16>(v, wstart, wlen, sh, cnt, tile_excl, tb, order, ctl, flags, epoch, nlists);
}

// ---- incremental snapshots (dirty tags) -----------------------------------------------------------------------------
// A snapshot (rollback point of a speculative epoch, or the user's snapshot()) is a second copy of items / leaf counts /
// node records that is kept in step with the live state by copying only what was written since it was last synchronised:
// every writer stamps the leaves / node records it modifies with the engine's serial (View::ldirty / vdirty), and an entry
// is dirty for a snapshot synchronised at serial S when its tag is > S.  to_live = 0: live -> snapshot ("commit": the
// snapshot catches up); to_live = 1: snapshot -> live ("rollback"), and the entry is re-tagged `newtag` so that the OTHER
// snapshot sees it as written.  One wave scans 64 tags per trip and copies the dirty leaves logN slots per lane group.
PMA_KERNEL void k_snap_sync_leaves(Edge *live, uint32_t *live_cnt, Edge *snap, uint32_t *snap_cnt, uint32_t *tag, uint64_t nleaves,
                                   int sh, uint32_t synced, uint32_t newtag, uint32_t to_live, unsigned long long *copied) {
  const int lane = wv::lane();
  const uint32_t logN = 1u << sh;
  const uint32_t G = logN >= 64u ? 1u : (64u >> sh);  // leaves copied per trip
  const uint64_t wstride = (uint64_t)wv::grid_dim() * (wv::block_dim() >> 6);
  unsigned long long mine = 0;
  for (uint64_t base = ((uint64_t)wv::block_idx() * (wv::block_dim() >> 6) + wv::wave_in_block()) * 64u; base < nleaves; base += wstride * 64u) {
    const uint64_t l = base + (uint64_t)lane;
    const uint32_t t = l < nleaves ? tag[l] : 0u;
    uint64_t m = wv::ballot(l < nleaves && t > synced);
    mine += (unsigned long long)wv::popc64(m);
    while (m) {
      uint64_t myleaf = ~0ull;
      for (uint32_t gI = 0; gI < G && m; gI++) {
        const int b = wv::ctz64(m);
        m &= m - 1ull;
        if (((uint32_t)lane >> sh) == gI || logN >= 64u) myleaf = base + (uint64_t)b;
      }
      if (myleaf != ~0ull) {
        const uint32_t q = (uint32_t)lane & (logN - 1u);
        for (uint32_t o = q; o < logN; o += 64u) {  // (logN <= 64: one trip)
          const uint64_t slot = (myleaf << sh) + o;
          if (to_live) live[slot] = snap[slot]; else snap[slot] = live[slot];
        }
        if (q == 0) {
          if (to_live) {
            live_cnt[myleaf] = snap_cnt[myleaf];
            tag[myleaf] = newtag;
          } else {
            snap_cnt[myleaf] = live_cnt[myleaf];
          }
        }
      }
    }
  }
  if (copied != nullptr && lane == 0 && mine) wv::atomic_add_u64(copied, mine);
}
PMA_KERNEL void k_snap_sync_nodes(Node *live, Node *snap, uint32_t *tag, uint64_t n, uint32_t synced, uint32_t newtag, uint32_t to_live) {
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t u = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); u < n; u += stride) {
    if (tag[u] > synced) {
      if (to_live) {
        live[u] = snap[u];
        tag[u] = newtag;
      } else {
        snap[u] = live[u];
      }
    }
  }
}
PMA_KERNEL void k_fill_u32(uint32_t *p, uint64_t n, uint32_t value) {
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t i = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); i < n; i += stride) p[i] = value;
}

// pppcsr_repartition: num_neighbors travels beside the edges (it is a counter of calls, not the degree: duplicate adds and
// deletes of missing edges move it, PCSR.cpp:1380/1409).  One record (vertex + base, num_neighbors, 1) per vertex out, and
// the setter for the records a partition receives (vertex partition-local again after the routing).
PMA_KERNEL void k_nn_export(View v, uint32_t base, Op *out) {
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t i = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); i < v.g.n; i += stride)
    out[i] = Op{(uint32_t)i + base, v.nodes[i].num_neighbors, 1u};
}
PMA_KERNEL void k_nn_set(View v, const Op *recs, uint64_t n) {
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t i = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); i < n; i += stride) {
    const Op r = recs[i];
    if (r.src < v.g.n) {
      v.nodes[r.src].num_neighbors = r.dst;
      v.vdirty[r.src] = v.serial;
    }
  }
}

PMA_KERNEL void k_copy_slots(const Edge *src, Edge *dst, uint64_t len) {
  const uint32_t *s = reinterpret_cast<const uint32_t *>(src);
  uint32_t *d = reinterpret_cast<uint32_t *>(dst);
  const uint64_t total = len * 3ull;
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t i = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); i < total; i += stride) d[i] = s[i];
}

// Are the vertex ranges still sorted, disjoint and consistent with nodes[]?  One wave per vertex: the node record
// (beginning / end chain, the sentinel on `beginning`) and every slot of (beginning, end): live slots carry src == vertex,
// are no sentinels and have strictly ascending dests.  Run after the one event that can break this (add_node after a
// doubling, PCSR.cpp:533-540 + 681-703): if nothing is wrong, the 64-ary search narrowing and the parallel rounds are valid again.
PMA_KERNEL void k_check_ranges(View v, unsigned long long *bad) {
  const int lane = wv::lane();
  const uint64_t wstride = (uint64_t)wv::grid_dim() * (wv::block_dim() >> 6);
  const uint32_t n = v.g.n;
  for (uint64_t u = (uint64_t)wv::block_idx() * (wv::block_dim() >> 6) + wv::wave_in_block(); u < n; u += wstride) {
    const Node nd = v.nodes[u];
    bool wrong = false;
    const uint64_t want_end = (u + 1 < n) ? (uint64_t)v.nodes[u + 1].beginning : v.g.N - 1;
    if ((uint64_t)nd.beginning >= v.g.N || (uint64_t)nd.end != want_end || nd.end <= nd.beginning) wrong = true;
    if (!wrong) {
      const Edge sn = v.items[nd.beginning];
      if (sn.src != (uint32_t)u || sn.dest != kMax || sn.value != (u == 0 ? kMax : (uint32_t)u)) wrong = true;
    }
    if (!wrong) {
      uint32_t prev = 0;
      bool have_prev = false;
      for (uint64_t base = (uint64_t)nd.beginning + 1; base < nd.end; base += 64) {
        const uint64_t s = base + (uint64_t)lane;
        Edge e = null_edge();
        if (s < nd.end) e = v.items[s];
        const bool live = s < nd.end && e.value != 0;
        const uint64_t m = wv::ballot(live);
        if (live && (e.src != (uint32_t)u || is_sentinel(e))) wrong = true;
        const uint64_t below = m & ((1ull << lane) - 1ull);
        const int pl = below ? 63 - __builtin_clzll(below) : 0;
        const uint32_t pd = wv::shfl(e.dest, pl);
        if (live && (below ? !(pd < e.dest) : (have_prev && !(prev < e.dest)))) wrong = true;
        if (m) {
          prev = wv::shfl(e.dest, 63 - __builtin_clzll(m));
          have_prev = true;
        }
      }
    }
    if (wv::ballot(wrong) != 0 && lane == 0) wv::atomic_add_u64(bad, 1ull);
  }
}

// ---- read-side kernels (get_neighbourhood PCSR.cpp:901-912, edge_exists :860-869) --------------------------
PMA_KERNEL void k_edge_exists(View v, uint32_t src, uint32_t dst, ExclOut *out) {
  dev::RangeRec rr;
  rr.plan = nullptr;
  rr.nr = 0;
  uint32_t found = 0;
  if (src < v.g.n) {
    const Node nd = v.nodes[src];
    dev::SearchHit hit_;
    const uint32_t loc = dev::pma_search(v, dst, nd.beginning + 1, nd.end, rr, &hit_);
    const Edge e = v.items[loc];
    found = (!is_null(e) && !is_sentinel(e) && e.dest == dst) ? 1u : 0u;
  }
  if (wv::lane() == 0) {
    out->found = found;
    out->result = X_DONE;
  }
}

// neighbours of one vertex: live dests in slots (beginning, end), in slot order; single workgroup of one wave
PMA_KERNEL void k_neighbourhood(View v, uint32_t src, int *outbuf, uint64_t cap, unsigned long long *count) {
  const int lane = wv::lane();
  unsigned long long run = 0;
  if (src < v.g.n) {
    const Node nd = v.nodes[src];
    for (uint64_t base = (uint64_t)nd.beginning + 1; base < (uint64_t)nd.end; base += 64) {
      const uint64_t s = base + (uint64_t)lane;
      Edge e = null_edge();
      if (s < (uint64_t)nd.end) e = v.items[s];
      const bool nn = e.value != 0;
      const uint64_t m = wv::ballot(nn);
      if (nn) {
        const unsigned long long o = run + dev::lanemask_lt_count(m, lane);
        if (outbuf && o < cap) outbuf[o] = (int)e.dest;
      }
      run += (unsigned long long)wv::popc64(m);
    }
  }
  if (lane == 0) *count = run;
}

// live edges per 64-slot chunk WITHOUT reading the edge array: leaf counts minus the sentinels that sit in the chunk
// (one atomic per vertex on a 4 B/chunk histogram), minus slot N-1 which is never part of a neighbourhood
PMA_KERNEL void k_chunk_sentinels(View v, uint32_t *chunk_sent) {
  // sentinel positions increase with the vertex id, so the sentinels of one chunk are a run of consecutive vertices:
  // the first vertex of each run counts the run and stores it (no atomics; chunks without sentinels stay 0).  Runs are
  // measured inside the wave with one ballot; the wave's last run may continue into the next 64 vertices (isolated
  // vertices sit shoulder to shoulder, up to 64 per chunk) and is finished with one more 64-wide probe.
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  const uint64_t n = v.g.n;
  const int lane = wv::lane();
  for (uint64_t base = (uint64_t)wv::block_idx() * wv::block_dim() + (wv::thread_idx() & ~63u); base < n; base += stride) {
    const uint64_t k = base + (uint64_t)lane;
    const bool valid = k < n;
    const uint32_t ch = valid ? (v.nodes[k].beginning >> 6) : kMax;
    uint32_t prev = wv::shfl(ch, lane == 0 ? 0 : lane - 1);
    if (lane == 0) prev = (k > 0) ? (v.nodes[k - 1].beginning >> 6) : kMax;
    const bool head = valid && (k == 0 || prev != ch);
    const uint64_t hm = wv::ballot(head);
    const int nvalid = wv::popc64(wv::ballot(valid));
    if (hm == 0) continue;  // the whole wave lies inside a run that an earlier wave counts
    const int lh = 63 - __builtin_clzll(hm);  // the wave's last run starts here
    if (head && lane != lh) {
      const uint64_t later = (hm >> (lane + 1)) << (lane + 1);
      chunk_sent[ch] = (uint32_t)(wv::ctz64(later) - lane);
    }
    const uint32_t chl = wv::shfl(ch, lh);
    uint32_t run = (uint32_t)(nvalid - lh);
    if (nvalid == 64) {  // a chunk holds at most 64 sentinels, so one probe of the next 64 vertices finishes the run
      const uint64_t k2 = base + 64 + (uint64_t)lane;
      const bool same = k2 < n && (v.nodes[k2].beginning >> 6) == chl;
      const uint64_t diff = wv::ballot(!same);
      run += diff ? (uint32_t)wv::ctz64(diff) : 64u;
    }
    if (lane == lh) chunk_sent[chl] = run;
  }
}
// live-edge count of every 64-slot chunk (leaf counts minus sentinels; slot N-1 is never part of a neighbourhood) and
// the sum over each tile of `tile_chunks` chunks; chunk_sent is left zeroed for the next scan
PMA_KERNEL void k_chunk_counts(View v, uint32_t *chunk_sent, uint32_t *chunkcnt, uint32_t tile_chunks, uint32_t *tilesum) {
  PMA_SHARED uint32_t red[4];
  const uint64_t N = v.g.N, nchunks = (N + 63) / 64;
  const uint32_t lpc = (v.g.logN >= 64) ? 1u : (64u >> v.g.sh);  // leaves per chunk
  const uint64_t ch = (uint64_t)wv::block_idx() * tile_chunks + wv::thread_idx();
  uint32_t c = 0;
  if (wv::thread_idx() < tile_chunks && ch < nchunks) {
    if (v.g.logN >= 64) {
      c = v.leafcnt[(ch * 64) >> v.g.sh];  // (logN = 64 only for N >= 2^32: not reachable, kept for completeness)
    } else {
      for (uint32_t q = 0; q < lpc; q++) {
        const uint64_t leaf = ch * lpc + q;
        if ((leaf << v.g.sh) < N) c += v.leafcnt[leaf];
      }
    }
    c -= chunk_sent[ch];
    chunk_sent[ch] = 0u;
    if (ch == nchunks - 1) {
      const Edge e = v.items[N - 1];
      if (e.value != 0 && !is_sentinel(e)) c -= 1u;
    }
    chunkcnt[ch] = c;
  }
  const uint32_t s = wv::reduce_add(c);
  if (wv::lane() == 0) red[wv::wave_in_block()] = s;
  wv::block_sync();
  if (wv::thread_idx() == 0) tilesum[wv::block_idx()] = red[0] + red[1] + red[2] + red[3];
}
// bulk neighbour scan (CSR export), final streaming pass: one workgroup per tile of chunks.  The tile's chunk counts are
// scanned in LDS (offset = scanned tile sum + in-tile prefix), then every wave streams its chunks — four in flight —
// writing dests in array order == CSR order and the row offsets at the sentinels.
// (contrib != nullptr: also emit, per edge, node_values[src] / num_neighbors(src) — the PageRank push of pagerank.h:21;
//  triples != nullptr: emit (src + src_base, dest, value) per edge instead of / besides dests)
PMA_KERNEL void k_scan_write(View v, const uint32_t *__restrict__ chunkcnt, uint32_t tile_chunks, const uint32_t *__restrict__ tile_excl,
                             unsigned long long *__restrict__ row_offsets, int *__restrict__ dests, uint64_t cap,
                             const float *__restrict__ node_values, float *__restrict__ contrib, Op *__restrict__ triples, uint32_t src_base) {
  PMA_SHARED uint32_t pre[256];
  PMA_SHARED uint32_t wsum[4];
  const int lane = wv::lane(), w = wv::wave_in_block();
  const uint64_t N = v.g.N;
  const uint64_t nchunks = (N + 63) / 64;
  const uint64_t tile = wv::block_idx();
  {
    const uint64_t ch = tile * tile_chunks + wv::thread_idx();
    const uint32_t x = (wv::thread_idx() < tile_chunks && ch < nchunks) ? chunkcnt[ch] : 0u;
    uint32_t incl = x;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = wv::shfl(incl, lane - o < 0 ? 0 : lane - o);
      if (lane >= o) incl += y;
    }
    if (lane == 63) wsum[w] = incl;
    wv::block_sync();
    uint32_t woff = 0;
    for (int q = 0; q < w; q++) woff += wsum[q];
    pre[wv::thread_idx()] = woff + incl - x;
  }
  wv::block_sync();
  const unsigned long long base = tile_excl[tile];
  const Edge *__restrict__ items = v.items;
  constexpr int K = 4;
  for (uint32_t c0 = (uint32_t)w * K; c0 < tile_chunks; c0 += 4 * K) {
    Edge e[K];
#pragma unroll
    for (int q = 0; q < K; q++) {
      const uint64_t s = (tile * tile_chunks + c0 + q) * 64 + (uint64_t)lane;
      e[q] = null_edge();
      if (c0 + q < tile_chunks && s < N) e[q] = items[s];
    }
#pragma unroll
    for (int q = 0; q < K; q++) {
      if (c0 + q >= tile_chunks) break;
      const uint64_t s = (tile * tile_chunks + c0 + q) * 64 + (uint64_t)lane;
      const bool nn = e[q].value != 0;
      const bool sent = nn && is_sentinel(e[q]);
      const bool live = nn && !sent && (s + 1 < N);
      const uint64_t m = wv::ballot(live);
      const unsigned long long o = base + pre[c0 + q] + dev::lanemask_lt_count(m, lane);
      if (live && o < cap && triples != nullptr)  // (pppcsr_repartition: the edge as an add of the global stream)
        triples[o] = Op{e[q].src + src_base, e[q].dest, e[q].value};
      if (live && o < cap && dests != nullptr) {
        dests[o] = (int)e[q].dest;
        if (contrib != nullptr) {
          if (e[q].dest >= v.g.n) dests[o] = (int)v.g.n;  // (the reference would write out of bounds; keeps the sort keys short)
          const uint32_t sv = e[q].src;
          contrib[o] = (sv < v.g.n) ? node_values[sv] / (float)v.nodes[sv].num_neighbors : 0.0f;
        }
      }
      if (sent && row_offsets != nullptr) {
        const uint32_t vid = (e[q].value == kMax) ? 0u : e[q].value;
        row_offsets[vid] = o;
      }
    }
  }
}

// ---- bulk build: an explicit NON-parity fast path ---------------------------------------------------
// The reference can only build a graph by single inserts, and the layout that produces is history dependent; this path
// builds a VALID packed-memory array (same invariants, same neighbourhoods, same num_neighbors) in a handful of passes:
// sort the adds by (src, dest) (stable: the last value of a duplicate wins, as it does when inserted one by one), then
// place sentinels and unique edges with the exact redistribute() positions of one whole-array window.
PMA_KERNEL void k_bb_keys(const Op *ops, uint64_t m, uint32_t n, unsigned long long *keys, uint32_t *vals) {
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t i = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); i < m; i += stride) {
    const Op o = ops[i];
    const bool ok = o.op != 0 && o.src < n;  // (add_edge ignores value 0 and src >= n, PCSR.cpp:1375-1377)
    keys[i] = ok ? (((unsigned long long)o.src << 32) | (unsigned long long)o.dst) : ((unsigned long long)n << 32);  // (sorts last)
    vals[i] = o.op;
  }
}
PMA_KERNEL void k_bb_flags(const unsigned long long *keys, uint64_t m, uint32_t n, uint32_t *flags) {
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t i = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); i < m; i += stride) {
    const unsigned long long k = keys[i];
    flags[i] = ((uint32_t)(k >> 32) < n && (i + 1 == m || keys[i + 1] != k)) ? 1u : 0u;  // last of its run = the value that survives
  }
}
PMA_DEV uint64_t bb_lower_bound(const unsigned long long *keys, uint64_t m, unsigned long long key) {
  uint64_t lo = 0, hi = m;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (keys[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// position of element k of the whole-array window, table in LDS, segment found by bisection
PMA_DEV uint64_t bb_pos(const ChainTable *tb, uint64_t k) {
  if (k == 0) return tb->index;
  const uint64_t t = tb->j - 1 - k;
  int lo = 0, hi = tb->nseg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tb->seg[mid].t0 <= t) lo = mid; else hi = mid - 1;
  }
  const ChainSeg &sg = tb->seg[lo];
  const uint64_t d = t - sg.t0;
  const uint64_t M = (d == 0) ? sg.M0 : (sg.M0 - sg.Dfirst - (d - 1) * sg.Drest);
  return M >> sg.shift;
}
PMA_DEV void bb_load_table(ChainTable *stb, const ChainTable *tb) {
  const uint32_t *g = reinterpret_cast<const uint32_t *>(tb);
  uint32_t *sp = reinterpret_cast<uint32_t *>(stb);
  const uint32_t words = (uint32_t)((sizeof(ChainTable) - sizeof(ChainSeg) * (size_t)(kMaxSeg - tb->nseg)) / 4);
  for (uint32_t i = wv::thread_idx(); i < words; i += wv::block_dim()) sp[i] = g[i];
  wv::block_sync();
}
// sentinel of every vertex + nodes[]: vertex u is element u + (unique edges of smaller sources) of the sequence
PMA_KERNEL void k_bb_vertices(View v, const unsigned long long *keys, uint64_t m, const uint32_t *rank, const unsigned long long *total,
                              const ChainTable *tb) {
  PMA_SHARED ChainTable stb;
  bb_load_table(&stb, tb);
  const uint64_t E = *total;
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t u = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); u < v.g.n; u += stride) {
    const uint64_t a = bb_lower_bound(keys, m, (unsigned long long)u << 32);
    const uint64_t b = bb_lower_bound(keys, m, (unsigned long long)(u + 1) << 32);
    const uint64_t before = (a < m) ? (uint64_t)rank[a] : E;  // unique edges whose source is smaller than u
    const uint64_t pos = bb_pos(&stb, u + before);
    Edge e;
    e.src = (uint32_t)u;
    e.dest = kMax;
    e.value = (u == 0) ? kMax : (uint32_t)u;
    v.items[pos] = e;
    v.nodes[u].beginning = (uint32_t)pos;
    v.nodes[u].num_neighbors = (uint32_t)(b - a);  // every add counts, duplicates included (PCSR.cpp:1408)
    if (u > 0) v.nodes[u - 1].end = (uint32_t)pos;
    if (u + 1 == v.g.n) v.nodes[u].end = (uint32_t)(v.g.N - 1);
  }
}
PMA_KERNEL void k_bb_edges(View v, const unsigned long long *keys, const uint32_t *vals, const uint32_t *flags, const uint32_t *rank,
                           uint64_t m, const ChainTable *tb) {
  PMA_SHARED ChainTable stb;
  bb_load_table(&stb, tb);
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t i = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); i < m; i += stride) {
    if (!flags[i]) continue;
    const unsigned long long k = keys[i];
    const uint32_t src = (uint32_t)(k >> 32);
    const uint64_t pos = bb_pos(&stb, (uint64_t)rank[i] + (uint64_t)src + 1ull);  // sentinels 0..src precede it
    Edge e;
    e.src = src;
    e.dest = (uint32_t)k;
    e.value = vals[i];
    v.items[pos] = e;
  }
}

// ---- graph-algorithm consumers over the gapped array (reference: src/utility/bfs.h, src/utility/pagerank.h) -----------
// BFS, one level per launch: one wave per frontier vertex walks its slot range (beginning, end) 64 slots at a time, skips
// nulls, claims unvisited neighbours with a compare-and-swap on their level and appends them to the next frontier (one
// atomic per wave per 64 slots).  Levels are unique, so the result equals the reference's queue-based walk exactly.
constexpr uint64_t kBfsWaveSlots = 4096;  // longest slot range one wave walks on its own
PMA_KERNEL void k_bfs_level(View v, const uint32_t *front, uint32_t nfront, uint32_t level, uint32_t *levels, uint32_t *next,
                            uint32_t *next_count) {
  const int lane = wv::lane();
  const uint64_t wstride = (uint64_t)wv::grid_dim() * (wv::block_dim() >> 6);
  for (uint64_t f = (uint64_t)wv::block_idx() * (wv::block_dim() >> 6) + wv::wave_in_block(); f < nfront; f += wstride) {
    const uint32_t u = front[f];
    const Node nd = v.nodes[u];
    if ((uint64_t)nd.end - (uint64_t)nd.beginning > kBfsWaveSlots) {  // a hub: leave it to one streaming pass (k_bfs_edges)
      if (lane == 0) next_count[1] = 1u;
      continue;
    }
    for (uint64_t base = (uint64_t)nd.beginning + 1; base < (uint64_t)nd.end; base += 64) {
      const uint64_t s = base + (uint64_t)lane;
      uint32_t val = 0, dst = 0;
      if (s < (uint64_t)nd.end) {
        val = v.items[s].value;
        dst = v.items[s].dest;
      }
      bool won = false;
      if (val != 0 && dst < v.g.n && levels[dst] == kMax) won = wv::atomic_cas_u32(&levels[dst], kMax, level + 1u) == kMax;
      const uint64_t m = wv::ballot(won);
      if (m) {
        uint32_t b = 0;
        if (lane == 0) b = wv::atomic_add_u32(next_count, (uint32_t)wv::popc64(m));
        b = wv::shfl(b, 0);
        if (won) next[b + dev::lanemask_lt_count(m, lane)] = dst;
      }
    }
  }
}
// BFS level for a LARGE frontier: one streaming pass over the gapped array instead of one wave per frontier vertex (whose
// hubs would serialise the level): every live edge whose source sits on the current level claims its destination.  All
// writers of a level store the same value, so plain stores suffice; `found` counts the claims (an upper bound is enough:
// it only steers the choice of the next level's kernel, and zero means "done").
PMA_KERNEL void k_bfs_edges(View v, uint32_t level, uint32_t *levels, uint32_t *found) {
  const int lane = wv::lane();
  const uint64_t N = v.g.N, nchunks = (N + 63) / 64;
  const uint64_t wstride = (uint64_t)wv::grid_dim() * (wv::block_dim() >> 6);
  uint32_t mine = 0;
  for (uint64_t ch = (uint64_t)wv::block_idx() * (wv::block_dim() >> 6) + wv::wave_in_block(); ch < nchunks; ch += wstride) {
    const uint64_t s = ch * 64 + (uint64_t)lane;
    Edge e = null_edge();
    if (s + 1 < N) e = v.items[s];  // (slot N-1 is never part of a neighbourhood)
    const bool live = e.value != 0 && !is_sentinel(e) && e.src < v.g.n && e.dest < v.g.n;
    if (live && levels[e.src] == level && levels[e.dest] == kMax) {
      levels[e.dest] = level + 1u;
      mine++;
    }
  }
  mine = wv::reduce_add(mine);
  if (lane == 0 && mine) wv::atomic_add_u32(found, mine);
}
// The streaming level, bitmap form.  The level's two per-edge tests — "is the source on the frontier", "is the destination
// still unvisited" — used to be two gathers from levels[] (4 MB at n = 1 M: 64-B lines fetched for 4 B, and the pass ran
// at 0.7-1.3 TB/s).  k_bfs_bits packs both answers into two bitmaps of n/8 bytes (128 KB: L2-resident on every XCD) with
// one coalesced sweep over levels[] per level; k_bfs_edges_bits then streams the array with four 64-slot chunks in flight
// per wave and touches levels[] only for edges into vertices that were unvisited when the level began.  Measured on an
// RMAT-20 / 10 M-edge graph (201 MB of slots): 35-38 us on light levels (5.5 TB/s), 63 / 40 us on the two heavy ones.
// (Claiming destinations with atomic ORs into the visited bitmap instead — exact `found`, one store per vertex — cost
// 195 / 100 us: 0.4 M atomics on 1024 cache lines are served by the memory side one line at a time.)
PMA_KERNEL void k_bfs_bits(const uint32_t *levels, uint32_t n, uint32_t level, uint32_t *front_bits, uint32_t *visited_bits) {
  const int lane = wv::lane();
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t base = (uint64_t)wv::block_idx() * wv::block_dim() + (wv::thread_idx() & ~63u); base < n; base += stride) {
    const uint64_t u = base + (uint64_t)lane;
    const uint32_t lv = u < n ? levels[u] : kMax;
    const uint64_t mf = wv::ballot(u < n && lv == level), mv = wv::ballot(u < n && lv != kMax);
    if (lane < 2) {
      front_bits[(base >> 5) + lane] = (uint32_t)(mf >> (32 * lane));
      visited_bits[(base >> 5) + lane] = (uint32_t)(mv >> (32 * lane));
    }
  }
}
constexpr uint32_t kBfsStripes = 64, kBfsStripeWords = 32;
PMA_KERNEL void k_bfs_edges_bits(View v, uint32_t level, const uint32_t *__restrict__ front_bits, const uint32_t *__restrict__ visited_bits,
                                 uint32_t *levels, uint32_t *found) {
  const int lane = wv::lane();
  const uint64_t N = v.g.N, nchunks = (N + 63) / 64;
  const uint64_t wstride = (uint64_t)wv::grid_dim() * (wv::block_dim() >> 6);
  const uint32_t n = v.g.n;
  uint32_t mine = 0;
  constexpr int kB = 4;
  for (uint64_t ch0 = ((uint64_t)wv::block_idx() * (wv::block_dim() >> 6) + wv::wave_in_block()) * kB; ch0 < nchunks; ch0 += wstride * kB) {
    Edge e[kB];
#pragma unroll
    for (int b = 0; b < kB; b++) {
      const uint64_t s = (ch0 + b) * 64 + (uint64_t)lane;
      e[b] = null_edge();
      if (s + 1 < N) e[b] = v.items[s];  // (slot N-1 is never part of a neighbourhood)
    }
    // Four phases, each over all kB chunks, so that the kB gathers of a phase are in flight TOGETHER (written one chunk after
    // the other, the levels[] load of chunk b+1 waits for the store of chunk b: they may alias).
    bool hit[kB];
    uint32_t bit[kB], old[kB];
#pragma unroll
    for (int b = 0; b < kB; b++) {
      const bool live = e[b].value != 0 && !is_sentinel(e[b]) && e[b].src < n && e[b].dest < n;
      hit[b] = live && ((front_bits[e[b].src >> 5] >> (e[b].src & 31u)) & 1u);
      bit[b] = 1u << (e[b].dest & 31u);
    }
#pragma unroll
    for (int b = 0; b < kB; b++) old[b] = hit[b] ? visited_bits[e[b].dest >> 5] : 0xFFFFFFFFu;
    // (the bitmap is the state at the start of the level; a look at levels[] itself — only for edges into NEW vertices —
    // keeps most of the repeated stores away.  It may be stale: all writers of a level store the same value.)
#pragma unroll
    for (int b = 0; b < kB; b++) {
      hit[b] = (old[b] & bit[b]) == 0u;
      old[b] = hit[b] ? levels[e[b].dest] : 0u;
    }
#pragma unroll
    for (int b = 0; b < kB; b++) {
      if (hit[b] && old[b] == kMax) {
        levels[e[b].dest] = level + 1u;
        mine++;
      }
    }
  }
  // (`found` is kBfsStripes counters on cache lines of their own, one add per workgroup: on a heavy level nearly every wave
  // has claims, and 32 K adds to ONE word are served one after the other by the memory side — that was 260-290 us of the
  // 320 / 290 us the heavy levels took, whatever the per-edge work looked like)
  PMA_SHARED uint32_t red[4];
  mine = wv::reduce_add(mine);
  if (lane == 0) red[wv::wave_in_block()] = mine;
  wv::block_sync();
  if (wv::thread_idx() == 0) {
    const uint32_t all = red[0] + red[1] + red[2] + red[3];
    if (all) wv::atomic_add_u32(found + (uint64_t)(wv::block_idx() % kBfsStripes) * kBfsStripeWords, all);
  }
}
// frontier list of one level (used when a small frontier follows an edge-centric level)
PMA_KERNEL void k_bfs_collect(const uint32_t *levels, uint32_t n, uint32_t level, uint32_t *front, uint32_t *count) {
  const int lane = wv::lane();
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t base = (uint64_t)wv::block_idx() * wv::block_dim() + (wv::thread_idx() & ~63u); base < n; base += stride) {
    const uint64_t u = base + (uint64_t)lane;
    const bool in = u < n && levels[u] == level;
    const uint64_t m = wv::ballot(in);
    if (m) {
      uint32_t b = 0;
      if (lane == 0) b = wv::atomic_add_u32(count, (uint32_t)wv::popc64(m));
      b = wv::shfl(b, 0);
      if (in) front[b + dev::lanemask_lt_count(m, lane)] = (uint32_t)u;
    }
  }
}
// PageRank push, last step: contributions sorted (stably) by destination; every destination's run is added IN ORDER —
// ascending source, the order in which the reference's loop adds them — so the fp32 sums are the reference's bit for bit.
// One thread per destination handles short runs; a run of kPrLongRun or more is queued for k_pr_longruns.
constexpr uint32_t kPrLongRun = 128;
PMA_KERNEL void k_pr_segsum(const uint32_t *__restrict__ keys, const float *__restrict__ vals, uint64_t m, uint32_t n, float *out,
                            uint32_t *long_list, uint32_t *long_count) {
  const uint64_t stride = (uint64_t)wv::grid_dim() * wv::block_dim();
  for (uint64_t d = (uint64_t)wv::block_idx() * wv::block_dim() + wv::thread_idx(); d < n; d += stride) {
    uint64_t lo = 0, hi = m;  // first position with keys[pos] >= d
    while (lo < hi) {
      const uint64_t mid = (lo + hi) >> 1;
      if (keys[mid] < (uint32_t)d) lo = mid + 1; else hi = mid;
    }
    uint64_t lo2 = lo, hi2 = (lo + kPrLongRun < m) ? lo + kPrLongRun : m;  // first position (within reach) with keys[pos] > d
    while (lo2 < hi2) {
      const uint64_t mid = (lo2 + hi2) >> 1;
      if (keys[mid] <= (uint32_t)d) lo2 = mid + 1; else hi2 = mid;
    }
    if (lo2 - lo >= kPrLongRun) {  // long (or longer) run: a whole wave streams it
      long_list[wv::atomic_add_u32(long_count, 1u)] = (uint32_t)d;
      continue;
    }
    float acc = 0.0f;
    for (uint64_t i = lo; i < lo2; i++) acc += vals[i];
    out[d] = acc;
  }
}
// one wave per long run: 64 contributions are loaded at once, then added one after the other in order (the adds are the
// serial part by definition; the loads no longer are)
PMA_KERNEL void k_pr_longruns(const uint32_t *__restrict__ keys, const float *__restrict__ vals, uint64_t m, const uint32_t *long_list,
                              const uint32_t *long_count, float *out) {
  const int lane = wv::lane();
  const uint32_t nl = *long_count;
  const uint64_t wstride = (uint64_t)wv::grid_dim() * (wv::block_dim() >> 6);
  for (uint64_t w = (uint64_t)wv::block_idx() * (wv::block_dim() >> 6) + wv::wave_in_block(); w < nl; w += wstride) {
    const uint32_t d = long_list[w];
    uint64_t lo = 0, hi = m;
    while (lo < hi) {
      const uint64_t mid = (lo + hi) >> 1;
      if (keys[mid] < d) lo = mid + 1; else hi = mid;
    }
    float acc = 0.0f;
    for (uint64_t base = lo; base < m; base += 64) {
      const uint64_t i = base + (uint64_t)lane;
      const bool in = i < m && keys[i] == d;
      float x = 0.0f;
      if (in) x = vals[i];
      const uint64_t mm = wv::ballot(in);
      const int cnt = wv::popc64(mm);  // (the run is contiguous: lanes 0 .. cnt-1)
      for (int q = 0; q < cnt; q++) acc += wv::shfl_f32(x, q);
      if (cnt < 64) break;
    }
    if (lane == 0) out[d] = acc;
  }
}

// =====================================================================================================================
// Speculative rounds ("optimistic mode"): commit more than a strict prefix per round, validated, with rollback.
//
// A round plans the M lowest pending updates (the carry list of deferred updates, then fresh ones from the stream).
// Every plan reserves its write leaves (wres) and read leaves (rres) with atomicMin(stream index).  An update PASSES
// when no earlier pending update writes anything it reads or writes and no earlier pending update reads anything it
// writes.  Passing updates commit unless an earlier update of the same REGION (aligned block of 2^regshift leaves)
// failed this round — a per-region strict prefix, which keeps later updates from overtaking a deferred update inside
// the block where its footprint can still move (rebalance windows are aligned power-of-two blocks no larger than a
// region, so a deferred update's window cannot leave its region).
//
// Soundness does not rest on that heuristic: every committed update stamps the leaves it read (rstamp) and wrote
// (wstamp) with its stream index, and an update may only commit if no LATER update has already written a leaf it
// reads or writes, nor read a leaf it writes.  With that check the executed schedule is conflict-serialisable in
// stream order (every conflicting pair ran in index order), i.e. identical to the reference's sequential result.  A
// failed check raises `violation`: the host restores the epoch snapshot and replays the epoch with the strict prefix
// rounds above.  K_EXCL updates are barriers: nothing later commits until the exclusive executor has run them.
// =====================================================================================================================
struct OptCtl {
  uint32_t carry_n[2], next_fresh[2], hor[2];
  uint32_t e1;  // end of the epoch (exclusive stream index)
  uint32_t violation, excl, done, error;
  uint32_t max_horizon, excl_idx;  // max_horizon: width of the launched grid (the next round's horizon never exceeds it)
  uint32_t width_cap;              // upper bound of the adaptive width (the engine's opt_horizon)
  uint32_t resident;               // waves the chip holds at once (0 = unknown): above it the width moves in whole multiples
  uint32_t maxc;  // 1 + largest stream index committed in this epoch
  uint32_t viol_idx;  // smallest stream index whose commit-time validation failed
  uint32_t adaptive;     // 1: adapt cur_horizon to the share of a round that commits (see compact_block)
  uint32_t cur_horizon;  // adaptive round width (<= max_horizon): grows while most of the round commits, shrinks otherwise
  unsigned long long gbar[2];  // keyed min index of a K_EXCL update in the horizon
  unsigned long long sbar[2];  // keyed min index of a SOFT barrier (a planned window close to the exclusive threshold): a word of
                               // its own — folded into gbar as key + 1 it was indistinguishable from a real K_EXCL key of update
                               // idx + 1, and o_compact then sent that update to the exclusive executor whatever its kind
  unsigned long long rounds, committed, planned, blocked, failed;
  uint32_t viol_info[8];  // debug: kind, leaf, stamp, what(1=wstamp on W,2=rstamp on W,3=wstamp on R), wleaf_lo, wleaf_hi, index, round
  uint32_t hist[192];  // debug: (horizon << 16 | committed) >> of the first rounds of the epoch
  // soft barriers with an extent ("zones"): an update whose planned window is close to the exclusive threshold keeps LATER
  // updates out of the aligned block its window may still grow into — not out of the whole array (a config #4 partition at
  // critical density had 90 % of its non-commits "behind a barrier" that sat megabytes away from them)
  uint32_t nzones[2];
  uint32_t zone_lo[2][8], zone_hi[2][8];  // inclusive leaf range
  unsigned long long zone_key[2][8];
  // diagnostics (option "diag"): why planned updates did not commit, first reason found per update
  // 0 exclusive kind, 1 behind a barrier (gbar), 2 duplicate-slot conflicts, 3 write leaf reserved by an earlier writer,
  // 4 write leaf read by an earlier update, 5 read leaf written by an earlier update, 6 sentinel located by is moved earlier,
  // 7 sentinel we move is needed earlier, 8 region prefix, 9 growth zone of a deferred reader/writer (pfail), 10 stamp violation
  unsigned long long why[12];
  // (diag) why chains ended: 0 list exhausted, 1 foreign / barrier / overflow stop index, 2 step limit, 3 exclusive kind, 4 read ranges
  // beyond the register copy, 5 footprint leaves the region, 6 window for a workgroup, 7 stamps, 8 region under a queued big window,
  // 9 heads, 10 chain steps
  uint32_t bk_round;  // the round whose bucket offsets OptArgs::bk_base holds (0: none)
  uint32_t nown[2][8];  // buckets with something to chain this round (OptArgs::owners: 8 sub-lists, so that the appends do not
                        // all hit one counter), by round parity
  unsigned long long chain_why[12];
  unsigned long long chain_stop[8];  // (diag) which stop index: 1 exclusive, 2 global soft barrier, 3 foreign update (xmin), 4 list overflow, 5 zone
  uint32_t njobs[2];  // big-window rebalances queued by this round's o_apply (by round parity; the next round's entry is reset by o_compact)
  uint32_t jobs_round[2];  // the round that queued them (launches that follow an exclusive / final round must not run them again)
  uint32_t skip;   // stream index the exclusive executor has just run inside this epoch (kMax: none); its slot commits as nothing
  uint32_t resume_par;  // round parity whose double-buffered entries (hor / carry_n / next_fresh / carry list) are current: the
                        // launches queued behind an exclusive update return at once and do not flip them
};
struct OptArgs {
  View v;
  const Op *ops;
  Plan *plans;
  uint32_t *opidx, *status, *vdbg;
  uint32_t *carry0, *carry1;
  OptCtl *ctl;
  StatShard *stats;
  unsigned long long *regfail;
  unsigned long long *pfail;  // per-leaf: smallest deferred update whose footprint may still grow over this leaf
  uint32_t *wstamp, *rstamp;
  uint32_t *vws, *vrs;  // per vertex: 1 + latest committed update that moved / read the position of its sentinel
  uint32_t round;
  int regshift;
  uint32_t diag;
  uint32_t defer_barrier;  // 0: off
  uint32_t soft_barrier;   // slots: see o_plan
  uint32_t zone_factor;    // 0: a soft barrier holds back every later update; f > 0: only those that touch the aligned block of
                           // f x its planned window (OptCtl::zone_*)
  // windows above big_min slots are rebalanced by a workgroup of o_big (job queue + one scratch stretch per workgroup)
  uint32_t big_min;
  dev::BigJob *jobs;
  Edge *bigscratch;
  uint32_t bigscratch_stride;  // slots per workgroup
  // in-round chains (o_chain): the planned updates of a round listed per region.  Regions hash into kChainBuckets buckets;
  // o_plan counts (its arrival number in the bucket is the update's place in the bucket's list), o_bscan turns the counts into
  // offsets, o_check writes the horizon slot into its place; after o_apply the wave of the bucket's FIRST arrival sorts the
  // list by horizon slot (= stream order) and executes what is still pending, one update after the other
  uint32_t chain;              // 0: off; k: at most k chained updates per bucket and round
  uint32_t chain_fence;        // (experiments: extra fences between chained updates, see wv::fence_mode)
  int chshift;                 // chain regions: 2^chshift leaves (1024 slots whatever the region rule's width is)
  unsigned long long *bk_cnt;  // [kChainBuckets] (round << 32) | updates listed this round
  uint32_t *bk_base;           // [kChainBuckets] offset of the bucket's list in bk_list
  uint32_t *bk_list;           // horizon slots, bucket after bucket
  uint32_t *bk_pos;            // per horizon slot: arrival number in its bucket (kMax: not listed)
  uint32_t *bk_reg;            // per horizon slot: region of the target slot
  uint32_t *owners;            // 8 sub-lists of owners_cap entries: a horizon slot of every bucket that has something to chain
  uint32_t owners_cap;
  uint32_t *bk_flag;           // [kChainBuckets] the round in which the bucket entered the work list
  unsigned long long *xmin;    // per region: keyed min stream index of a PENDING update that touches the region but lives elsewhere
  // diag >= 2: per update of the batch {rounds it failed in, code of the last failure, stream index of what blocked it,
  // planned window}: the dependency chains of an epoch can be followed afterwards (tools/diag_chains.py)
  uint32_t *dg;
};
constexpr uint32_t kChainBuckets = 1u << 15;
constexpr uint32_t kChainRegionSlots = 1024;  // chain regions are at most this many slots (12 KB of LDS)
constexpr uint32_t kChainList = 256;  // a wave sorts this many listed updates of one bucket in its LDS tile; longer lists run without a chain
PMA_DEV uint32_t chain_bucket(uint32_t region) { return (region * 0x9E3779B1u) >> 17; }
constexpr uint32_t kBigJobs = 256;  // capacity of the round's job queue
constexpr uint32_t kMaxZones = 8;   // soft-barrier zones per round; the ninth barrier of a round is a global one
constexpr uint32_t kRegionPadLeaves = 2u;
constexpr uint32_t kGrowLeaves = 8u;
constexpr uint32_t OS_PASS = 1u, OS_STAMP_BAD = 2u, OS_COMMITTED = 4u;
// committed by a chain (o_chain): for the compaction as good as OS_COMMITTED, but NOT for the waves of o_chain that look for
// the head of their region while the chain is running — they must keep seeing the state the launch started with
constexpr uint32_t OS_CHAINED = OS_COMMITTED | 8u;


PMA_DEV bool key_earlier(unsigned long long k, uint32_t tag, uint32_t idx) { return (uint32_t)(k >> 32) == tag && (uint32_t)k < idx; }

// The plan record's header and this lane's read range, requested in ONE batch (and before the kernel's early-exit tests:
// the per-wave arrays are padded to the launch grid, so the loads are always in bounds).  The round kernels are chains
// of dependent loads; what can be asked for together is asked for together.
constexpr int kHeadRanges = 8;
struct PlanHead {
  uint32_t kind, index, wstart, wlen, wleaf_lo, wleaf_hi, mv_lo, mv_hi, sleaf_b, sleaf_e, nr, nlong, sdep;
  uint32_t my_lo, my_hi;  // lane r: read range r (r < 64)
};
PMA_DEV PlanHead load_plan_head(const Plan *pl, int lane) {
  PlanHead h;
  h.kind = pl->kind;
  h.index = pl->index;
  h.wstart = pl->wstart;
  h.wlen = pl->wlen;
  h.wleaf_lo = pl->wleaf_lo;
  h.wleaf_hi = pl->wleaf_hi;
  h.mv_lo = pl->mv_lo;
  h.mv_hi = pl->mv_hi;
  h.sleaf_b = pl->sleaf_b;
  h.sleaf_e = pl->sleaf_e;
  h.nr = pl->nr;
  h.nlong = pl->nlong;
  h.sdep = pl->sdep;
  // (the first kHeadRanges ranges only: an update records two or three — its search certificate and the leaves of a short
  // climb — and 64 lanes x 2 words fetched 512 B of a 640-B record for nothing; longer lists are walked from the record)
  h.my_lo = lane < kHeadRanges ? pl->rlo[lane] : 1u;
  h.my_hi = lane < kHeadRanges ? pl->rhi[lane] : 0u;
  return h;
}
#define PMA_FOR_EACH_READ_LEAF_H(h, pl, lane, LEAFVAR, BODY)                                  \
  do {                                                                                        \
    if ((h).nr <= (uint32_t)kHeadRanges && (h).nlong == 0u) {                                 \
      if ((uint32_t)(lane) < (h).nr)                                                          \
        for (uint32_t LEAFVAR = (h).my_lo; LEAFVAR <= (h).my_hi; LEAFVAR++) { BODY; }         \
    } else {                                                                                  \
      PMA_FOR_EACH_READ_LEAF(pl, lane, LEAFVAR, BODY);                                        \
    }                                                                                         \
  } while (0)

template <bool EXTRAS>
PMA_DEV void o_plan_t(const OptArgs &a) {
  OptCtl *c = a.ctl;
  const uint32_t par = a.round & 1u;
  const uint32_t wid = wv::block_idx() * (wv::block_dim() >> 6) + (uint32_t)wv::wave_in_block();
  const uint32_t *carry = par ? a.carry1 : a.carry0;
  const uint32_t f_done = c->done, f_viol = c->violation, f_excl = c->excl, f_err = c->error, f_skip = c->skip;
  const uint32_t hor = c->hor[par], cn = c->carry_n[par], nf = c->next_fresh[par];
  const uint32_t cw = carry[wid];  // (requested with the control block; the carry lists are padded to the launch grid)
  if (f_done || f_viol || f_excl || f_err) return;
  if (wid >= hor) return;
  const uint32_t used = cn < hor ? cn : hor;
  const uint32_t idx = (wid < used) ? cw : nf + (wid - used);
  const Op op = a.ops[idx];
  Plan *pl = &a.plans[wid];
  const int lane = wv::lane();
  if (idx == f_skip) {  // executed by the exclusive executor in the middle of this epoch: nothing left to do, commits at once
    if (lane == 0) {
      a.opidx[wid] = idx;
      pl->kind = K_SKIP;
      pl->index = pl->wstart = pl->wlen = pl->nr = pl->nlong = pl->sdep = 0;
      pl->wleaf_lo = 1;
      pl->wleaf_hi = 0;
      pl->mv_lo = 1;
      pl->mv_hi = 0;
      pl->sleaf_b = pl->sleaf_e = 0;
      if ((EXTRAS && a.chain)) a.bk_pos[wid] = kMax;
    }
    return;
  }
  // the plan record goes to memory for o_check / o_apply; this kernel reserves straight from the registers
  const dev::PlanRegs pr = dev::plan_op(a.v, op, pl);
  if (lane == 0) a.opidx[wid] = idx;
  const unsigned long long key = make_key(a.round, idx);
  const uint32_t kind = pr.kind;
  if (kind == K_EXCL) {
    if (lane == 0) wv::atomic_min_u64(&c->gbar[par], key);
    if (lane == 0 && (EXTRAS && a.chain)) a.bk_pos[wid] = kMax;
    return;
  }
  if (kind == K_DUP) {
    if (lane == 0) wv::atomic_min_u64(&a.v.dres[pr.wleaf_lo], key);
  } else if (kind_strong(kind)) {
    const uint32_t wl = pr.wleaf_lo, wh = pr.wleaf_hi;
    for (uint32_t leaf = wl + (uint32_t)lane; leaf <= wh; leaf += 64) wv::atomic_min_u64(&a.v.wres[leaf], key);
    // an update whose window is already within two levels of the exclusive threshold is likely to turn exclusive
    // once the earlier updates have landed: nothing later may overtake it (soft barrier)
    // ... and so is an update that has ALREADY been deferred at least once and whose window is big (a.defer_barrier
    // slots): its window keeps growing while it waits behind a hot range, and everything committed around it meanwhile
    // is a candidate for a rollback
    if ((pr.wlen >= a.soft_barrier || (a.defer_barrier && wid < used && pr.wlen >= a.defer_barrier)) && lane == 0) {
      uint32_t zslot = kMaxZones;
      if ((EXTRAS ? a.zone_factor : 0u)) zslot = wv::atomic_add_u32(&c->nzones[par], 1u);
      if (zslot < kMaxZones) {
        const uint32_t nleaves = (uint32_t)(a.v.g.N >> a.v.g.sh);
        uint64_t zl = (uint64_t)(pr.wlen >> a.v.g.sh) * (EXTRAS ? a.zone_factor : 0u);  // leaves in the block (a power of two when the factor is)
        if (zl > nleaves) zl = nleaves;
        uint64_t blo = (uint64_t)(pr.wstart >> a.v.g.sh) / zl * zl, bhi = blo + zl - 1u;
        if (wl < blo) blo = wl;
        if (wh > bhi) bhi = wh;
        if (bhi >= nleaves) bhi = nleaves - 1u;
        c->zone_lo[par][zslot] = (uint32_t)blo;
        c->zone_hi[par][zslot] = (uint32_t)bhi;
        c->zone_key[par][zslot] = key;
      } else {
        wv::atomic_min_u64(&c->sbar[par], key);
      }
    }
    const uint32_t ml = pr.mv_lo, mh = pr.mv_hi;
    for (uint64_t u = (uint64_t)ml + (uint64_t)lane; u <= (uint64_t)mh && ml <= mh; u += 64) wv::atomic_min_u64(&a.v.vw[u], key);
  }
  if (pr.nr <= 64u && pr.nlong == 0u) {  // lane r holds read range r
    if ((uint32_t)lane < pr.nr)
      for (uint32_t leaf = pr.my_lo; leaf <= pr.my_hi; leaf++) wv::atomic_min_u64(&a.v.rres[leaf], key);
  } else {
    wv::fence();  // (rare) more ranges than lanes, or long ranges: walk the record this wave has just written
    PMA_FOR_EACH_READ_LEAF(pl, lane, leaf, wv::atomic_min_u64(&a.v.rres[leaf], key));
  }
  if (kind_real(kind) && op.src < a.v.g.n) {  // readers of the positions of sentinels src and src+1 (only when the result depends on them)
    if (lane == 0 && (pr.sdep & 1u)) wv::atomic_min_u64(&a.v.vr[op.src], key);
    if (lane == 1 && (pr.sdep & 2u) && op.src + 1u < a.v.g.n) wv::atomic_min_u64(&a.v.vr[op.src + 1u], key);
  }
  if ((EXTRAS && a.chain) && lane == 0) {  // list the update under the region of its target slot (arrival order: o_chain sorts)
    uint32_t region = kMax, pos = kMax;
    if (kind_real(kind)) {
      region = (pr.index >> a.v.g.sh) >> a.chshift;
      const uint32_t b = chain_bucket(region);
      wv::atomic_max_u64(&a.bk_cnt[b], (unsigned long long)a.round << 32);  // (first of the round: the count restarts at 0)
      pos = (uint32_t)wv::atomic_add_u64(&a.bk_cnt[b], 1ull);
    }
    a.bk_pos[wid] = pos;
    a.bk_reg[wid] = region;
  }
}
PMA_KERNEL void o_plan(OptArgs a) { o_plan_t<false>(a); }
PMA_KERNEL void o_plan_x(OptArgs a) { o_plan_t<true>(a); }

// A deferred update tells the regions it touches WITHOUT living there (its window, slide pad, growth block, read ranges or
// the sentinels it locates its range by reach into them) that somebody earlier is still pending: a chain of that region
// stops before overtaking it.  [ulo, uhi]: the union of its write range (padded) and growth block, in leaves.
PMA_DEV void mark_foreign_regions(const OptArgs &a, const PlanHead &h, const Plan *pl, unsigned long long key, uint32_t ulo, uint32_t uhi, int lane) {
  const uint32_t mine = (h.index >> a.v.g.sh) >> a.chshift;
  for (uint32_t g = (ulo >> a.chshift) + (uint32_t)lane; g <= (uhi >> a.chshift); g += 64)
    if (g != mine) wv::atomic_min_u64(&a.xmin[g], key);
  PMA_FOR_EACH_READ_LEAF_H(h, pl, lane, leaf, { if ((leaf >> a.chshift) != mine) wv::atomic_min_u64(&a.xmin[leaf >> a.chshift], key); });
  if (lane == 0 && (h.sdep & 1u) && (h.sleaf_b >> a.chshift) != mine) wv::atomic_min_u64(&a.xmin[h.sleaf_b >> a.chshift], key);
  if (lane == 1 && (h.sdep & 2u) && (h.sleaf_e >> a.chshift) != mine) wv::atomic_min_u64(&a.xmin[h.sleaf_e >> a.chshift], key);
}

// o_check's part of the chain lists: the update's horizon slot goes to its place in the bucket's list, and the first update
// of a bucket that FAILS its check enters the bucket in the round's work list (a bucket whose updates all pass has nothing
// left to chain)
PMA_DEV void chain_list_entry(const OptArgs &a, OptCtl *c, uint32_t par, uint32_t wid, bool passed) {
  if (c->bk_round != a.round) return;
  const uint32_t pos = a.bk_pos[wid];
  if (pos == kMax) return;
  const uint32_t b = chain_bucket(a.bk_reg[wid]);
  a.bk_list[a.bk_base[b] + pos] = wid;
  if (!passed && wv::atomic_exch_u32(&a.bk_flag[b], a.round) != a.round) {
    const uint32_t sub = wid & 7u;
    const uint32_t at = wv::atomic_add_u32(&c->nown[par][sub], 1u);
    if (at < a.owners_cap) a.owners[(uint64_t)sub * a.owners_cap + at] = wid;
  }
}

template <bool EXTRAS>
PMA_DEV void o_check_t(const OptArgs &a) {
  OptCtl *c = a.ctl;
  const uint32_t par = a.round & 1u;
  const uint32_t wid = wv::block_idx() * (wv::block_dim() >> 6) + (uint32_t)wv::wave_in_block();
  const int lane = wv::lane();
  const uint32_t f_done = c->done, f_viol = c->violation, f_excl = c->excl, f_err = c->error;
  const uint32_t hor = c->hor[par];
  const unsigned long long gbar = c->gbar[par], sbar = c->sbar[par];
  const uint32_t nzones = c->nzones[par];  // (requested with the other control words: asked for where it is used, it is one more
                                           //  dependent round trip on every wave's path — 2 us per launch)
  const uint32_t idx = a.opidx[wid];
  const Plan *pl = &a.plans[wid];
  const PlanHead h = load_plan_head(pl, lane);
  if (f_done || f_viol || f_excl || f_err) return;
  if (wid >= hor) return;
  const uint32_t kind = h.kind;
  const unsigned long long key = make_key(a.round, idx);
  const uint32_t tag = (uint32_t)(key >> 32);
  // (a soft barrier holds back everything AFTER the update that raised it, that update itself may commit)
  bool fail = (kind == K_EXCL) || key_earlier(gbar, tag, idx) || key_earlier(sbar, tag, idx);
  uint32_t why = (kind == K_EXCL) ? 0u : (fail ? 1u : 99u);  // diagnostics: first reason (lowest code wins below)
  if (fail) {
    // An exclusive update, or one behind this round's barrier: it does not commit now, and neither does anything after it
    // (the barrier is earlier than all of them), so there is nobody to keep out of its regions and nothing to learn from
    // its footprint — which, for a climb towards the root, is every leaf of the array (a 3.5 ms walk by one wave, while
    // the rest of the launch waits).  Its stamps are looked at in the round that does check it.
    if ((EXTRAS && a.diag) && lane == 0) wv::atomic_add_u64(&c->why[why], 1ull);
    if (lane == 0 && (EXTRAS && a.diag) && a.dg != nullptr) {
      uint32_t *r = a.dg + 4ull * idx;
      r[0] += 1u;
      r[1] = why;
      r[2] = (kind == K_EXCL) ? idx : (uint32_t)(key_earlier(gbar, tag, idx) ? gbar : sbar);
      r[3] = h.wlen;
    }
    if (lane == 0) a.status[wid] = 0u;
    if ((EXTRAS && a.chain) && lane == 0) chain_list_entry(a, c, par, wid, false);
    return;
  }
#define PMA_WHY(code) do { if ((EXTRAS && a.diag) && (code) < why) why = (code); } while (0)
#define PMA_WHYB(code, bkey) do { if ((EXTRAS && a.diag) && (code) < why) { why = (code); blk = (uint32_t)(bkey); } } while (0)
  uint32_t blk = kMax;
  bool stamp_bad = false;
  const uint32_t me1 = idx + 1u;  // stamps hold (index + 1) of the latest committed toucher
  const bool writes = kind_writes(kind);
  const bool strong = kind_strong(kind);
  {  // zones of earlier soft-barrier updates: anything of ours inside one -> deferred (and treated like any other deferred
     // update below: it keeps later updates out of its own regions)
    uint32_t nz = EXTRAS ? nzones : 0u;
    if (nz > kMaxZones) nz = kMaxZones;
    for (uint32_t z = 0; z < nz; z++) {
      if (!key_earlier(c->zone_key[par][z], tag, idx)) continue;
      const uint32_t zlo = c->zone_lo[par][z], zhi = c->zone_hi[par][z];
      bool in = false;
      if (kind_real(kind)) {
        const uint32_t il = h.index >> a.v.g.sh;
        if (lane == 0) in = writes ? (h.wleaf_lo <= zhi && h.wleaf_hi >= zlo) : (il >= zlo && il <= zhi);
        if (lane == 1 && (h.sdep & 1u)) in = h.sleaf_b >= zlo && h.sleaf_b <= zhi;
        if (lane == 2 && (h.sdep & 2u)) in = h.sleaf_e >= zlo && h.sleaf_e <= zhi;
      }
      if (h.nr <= (uint32_t)kHeadRanges && h.nlong == 0u) {
        if ((uint32_t)lane < h.nr && h.my_lo <= zhi && h.my_hi >= zlo) in = true;
      } else {
        for (uint32_t r = (uint32_t)lane; r < h.nr; r += 64)
          if (pl->rlo[r] <= zhi && pl->rhi[r] >= zlo) in = true;
      }
      if (in) { fail = true; PMA_WHYB(1u, c->zone_key[par][z]); }
    }
  }
  if (kind == K_DUP) {
    const uint32_t leaf = h.wleaf_lo;
    if (key_earlier(a.v.wres[leaf], tag, idx)) { fail = true; PMA_WHY(2u); }  // an earlier pending update moves slots of this leaf
    if (a.v.dres[leaf] != key) { fail = true; PMA_WHY(2u); }                   // an earlier pending duplicate on this leaf
  }
  if (strong) {
    const uint32_t wl = h.wleaf_lo, wh = h.wleaf_hi;
    // (a big window spans thousands of leaves: four leaves per lane are requested together, 20 loads per trip)
    for (uint32_t base = wl; base <= wh; base += 256u) {
      unsigned long long kw[4], kd[4], kr[4];
      uint32_t sw[4], sr[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t leaf = base + (uint32_t)q * 64u + (uint32_t)lane;
        const bool in = leaf <= wh && leaf >= base;
        kw[q] = in ? a.v.wres[leaf] : key;
        kd[q] = in ? a.v.dres[leaf] : ~0ull;
        kr[q] = in ? a.v.rres[leaf] : ~0ull;
        sw[q] = in ? a.wstamp[leaf] : 0u;
        sr[q] = in ? a.rstamp[leaf] : 0u;
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t leaf = base + (uint32_t)q * 64u + (uint32_t)lane;
        if (kw[q] != key) { fail = true; PMA_WHYB(3u, kw[q]); }                   // an earlier pending update writes it
        if (key_earlier(kd[q], tag, idx)) { fail = true; PMA_WHYB(2u, kd[q]); }   // an earlier pending duplicate overwrites a slot here
        if (key_earlier(kr[q], tag, idx)) { fail = true; PMA_WHYB(4u, kr[q]); }   // an earlier pending update reads it
        if (sw[q] > me1 || sr[q] > me1) {  // a LATER update already touched it
          stamp_bad = true;
          a.vdbg[4 * wid + 0] = leaf;
          a.vdbg[4 * wid + 1] = sw[q] > me1 ? sw[q] : sr[q];
          a.vdbg[4 * wid + 2] = sw[q] > me1 ? 1u : 2u;
        }
      }
      if (wh - base < 256u) break;  // (no wrap-around at the top of the leaf range)
    }
  }
  PMA_FOR_EACH_READ_LEAF_H(h, pl, lane, leaf, {
    if (key_earlier(a.v.wres[leaf], tag, idx)) { fail = true; PMA_WHYB(5u, a.v.wres[leaf]); }  // an earlier pending update writes what we read
    if (a.wstamp[leaf] > me1) {                               // a LATER update already wrote what we read
      stamp_bad = true;
      a.vdbg[4 * wid + 0] = leaf;
      a.vdbg[4 * wid + 1] = a.wstamp[leaf];
      a.vdbg[4 * wid + 2] = 3u;
    }
  });
  if (kind_real(kind)) {
    const uint32_t src = a.ops[idx].src;
    if (src < a.v.g.n && lane < 2 && ((h.sdep >> lane) & 1u) && src + (uint32_t)lane < a.v.g.n) {  // lane 0: sentinel src, lane 1: sentinel src+1
      const uint32_t u = src + (uint32_t)lane;
      if (key_earlier(a.v.vw[u], tag, idx)) { fail = true; PMA_WHYB(6u, a.v.vw[u]); }  // an earlier pending update moves a sentinel we located by
      if (a.vws[u] > me1) {                                // a LATER update already moved it
        stamp_bad = true;
        a.vdbg[4 * wid + 0] = u;
        a.vdbg[4 * wid + 1] = a.vws[u];
        a.vdbg[4 * wid + 2] = 4u;
      }
    }
    if (strong) {
      const uint32_t ml = h.mv_lo, mh = h.mv_hi;
      for (uint64_t u = (uint64_t)ml + (uint64_t)lane; u <= (uint64_t)mh && ml <= mh; u += 64) {
        if (key_earlier(a.v.vr[u], tag, idx)) { fail = true; PMA_WHYB(7u, a.v.vr[u]); }  // an earlier pending update still needs the old position
        if (a.vrs[u] > me1 || a.vws[u] > me1) {              // a LATER update already used / moved it
          stamp_bad = true;
          a.vdbg[4 * wid + 0] = (uint32_t)u;
          a.vdbg[4 * wid + 1] = a.vrs[u] > me1 ? a.vrs[u] : a.vws[u];
          a.vdbg[4 * wid + 2] = 5u;
        }
      }
    }
  }
  const bool anyfail = wv::ballot(fail) != 0;
  const bool anybad = wv::ballot(stamp_bad) != 0;
  if ((EXTRAS && a.diag) && anyfail) {
    uint32_t w = why, wb = blk;
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t y = wv::shfl(w, lane ^ o), yb = wv::shfl(wb, lane ^ o);
      if (y < w || (y == w && yb < wb)) {
        w = y;
        wb = yb;
      }
    }
    if (lane == 0 && w < 12u) wv::atomic_add_u64(&c->why[w], 1ull);
    if (lane == 0 && a.dg != nullptr) {
      uint32_t *r = a.dg + 4ull * idx;
      r[0] += 1u;
      r[1] = w;
      r[2] = wb;
      r[3] = h.wlen;
    }
  }
  if (anyfail && kind_real(kind)) {
    // a deferred update keeps later updates out of its region(s); its footprint may still creep over a region edge
    // by a slide, so the mark is padded by kRegionPadLeaves leaves on both sides
    const uint32_t nleaves = (uint32_t)(a.v.g.N >> a.v.g.sh);
    uint32_t ll = writes ? h.wleaf_lo : (h.index >> a.v.g.sh), lh = writes ? h.wleaf_hi : ll;
    ll = (ll > kRegionPadLeaves) ? ll - kRegionPadLeaves : 0u;
    lh = (lh + kRegionPadLeaves < nleaves) ? lh + kRegionPadLeaves : nleaves - 1u;
    const uint32_t pglo = ll >> a.regshift, pghi = lh >> a.regshift;
    for (uint32_t g = pglo + (uint32_t)lane; g <= pghi; g += 64) wv::atomic_min_u64(&a.regfail[g], key);
    // leaf-level mark for later READERS: the deferred update's window can still grow to an ancestor block; cover
    // the aligned block of 4x its tentative window (at least kGrowLeaves leaves) plus the slide pad
    uint32_t wleaves = writes && h.wlen ? (h.wlen >> a.v.g.sh) : 1u;
    if (wleaves < 1u) wleaves = 1u;
    uint32_t blk = wleaves * 4u;
    if (blk < kGrowLeaves) blk = kGrowLeaves;
    const uint32_t anchor = writes && h.wlen ? (h.wstart >> a.v.g.sh) : (h.index >> a.v.g.sh);
    uint32_t bl = anchor & ~(blk - 1u), bh = bl + blk - 1u;
    if (ll < bl) bl = ll;
    if (lh > bh) bh = lh;
    if (bh >= nleaves) bh = nleaves - 1u;
    for (uint32_t leaf = bl + (uint32_t)lane; leaf <= bh; leaf += 64) wv::atomic_min_u64(&a.pfail[leaf], key);
    if ((EXTRAS && a.chain)) {  // (what it reserves NOW; its window may still grow — that is what validation is for)
      const uint32_t xl = writes ? h.wleaf_lo : (h.index >> a.v.g.sh), xh = writes ? h.wleaf_hi : xl;
      mark_foreign_regions(a, h, pl, key, xl, xh, lane);
    }
  }
  if (lane == 0) a.status[wid] = (anyfail ? 0u : OS_PASS) | (anybad ? OS_STAMP_BAD : 0u);
  if ((EXTRAS && a.chain) && lane == 0) chain_list_entry(a, c, par, wid, !anyfail);
}

PMA_KERNEL void o_check(OptArgs a) { o_check_t<false>(a); }
PMA_KERNEL void o_check_x(OptArgs a) { o_check_t<true>(a); }

// EXTRAS = false: the opt-in experiments (chains, zones) and the diagnostics are compiled out — carried along as run-time
// branches they cost the calm stream 4 % (config #2: 179 vs 187 M updates/s); the engine launches the *_x kernels when one is on
template <bool EXTRAS>
PMA_DEV void o_apply_wave(const OptArgs &a, uint32_t *lds_wave) {
  OptCtl *c = a.ctl;
  const uint32_t par = a.round & 1u;
  const uint32_t wid = wv::block_idx() * (wv::block_dim() >> 6) + (uint32_t)wv::wave_in_block();
  const int lane = wv::lane();
  const uint32_t f_done = c->done, f_viol = c->violation, f_excl = c->excl, f_err = c->error;
  const uint32_t hor = c->hor[par];
  const uint32_t st = a.status[wid];
  const uint32_t idx = a.opidx[wid];
  const Plan *pl = &a.plans[wid];
  const PlanHead h = load_plan_head(pl, lane);
  // the update itself: requested as soon as its index is known (unconditionally — slot 0 for waves beyond the horizon), so
  // that it travels while the region checks below wait for their own loads instead of after them
  const Op op = a.ops[(wid < hor) ? idx : 0u];
  if (f_done || f_viol || f_excl || f_err) return;
  if (wid >= hor) return;
  if (!(st & OS_PASS)) return;
  const uint32_t kind = h.kind;
  const unsigned long long key = make_key(a.round, idx);
  const uint32_t tag = (uint32_t)(key >> 32);
  const bool writes = kind_writes(kind);
  if (kind_real(kind)) {
    uint32_t glo, ghi;
    if (writes) {
      glo = h.wleaf_lo >> a.regshift;
      ghi = h.wleaf_hi >> a.regshift;
    } else {
      glo = ghi = (h.index >> a.v.g.sh) >> a.regshift;
    }
    bool blocked = false, blocked_r = false;
    uint32_t rblk = kMax;
    for (uint32_t g = glo + (uint32_t)lane; g <= ghi; g += 64)
      if (key_earlier(a.regfail[g], tag, idx)) {
        blocked_r = true;
        rblk = (uint32_t)a.regfail[g];
      }
    // ... nor may we have READ a leaf an earlier deferred update may still grow over
    PMA_FOR_EACH_READ_LEAF_H(h, pl, lane, leaf, { if (key_earlier(a.pfail[leaf], tag, idx)) blocked = true; });
    // ... nor located our range by a sentinel inside the block a deferred earlier update may still grow over
    if (lane == 0 && (h.sdep & 1u) && key_earlier(a.pfail[h.sleaf_b], tag, idx)) blocked = true;
    if (lane == 1 && (h.sdep & 2u) && key_earlier(a.pfail[h.sleaf_e], tag, idx)) blocked = true;
    const bool any_r = wv::ballot(blocked_r) != 0, any_p = wv::ballot(blocked) != 0;
    if ((EXTRAS && a.diag) && (any_r || any_p) && lane == 0) wv::atomic_add_u64(&c->why[any_r ? 8 : 9], 1ull);
    if ((EXTRAS && a.diag) && a.dg != nullptr && (any_r || any_p)) {
      for (int o = 32; o > 0; o >>= 1) {
        const uint32_t y = wv::shfl(rblk, lane ^ o);
        rblk = y < rblk ? y : rblk;
      }
      if (lane == 0) {
        uint32_t *r = a.dg + 4ull * idx;
        r[0] += 1u;
        r[1] = any_r ? 8u : 9u;
        r[2] = rblk;
        r[3] = h.wlen;
      }
    }
    if (any_r || any_p) {  // an earlier update of this region was deferred: keep stream order inside it
      if ((EXTRAS && a.chain)) {  // (still pending: chains of the other regions it touches must not overtake it)
        const uint32_t ll = writes ? h.wleaf_lo : (h.index >> a.v.g.sh), lh = writes ? h.wleaf_hi : ll;
        mark_foreign_regions(a, h, pl, key, ll, lh, lane);
      }
      return;
    }
  }
  if (st & OS_STAMP_BAD) {
    if ((EXTRAS && a.diag) && lane == 0) wv::atomic_add_u64(&c->why[10], 1ull);
    if (lane == 0) {
      const uint32_t prev = wv::atomic_min_u32(&c->viol_idx, idx);
      wv::atomic_exch_u32(&c->violation, 1u);
      if (idx < prev) {
        c->viol_info[0] = kind;
        c->viol_info[1] = a.vdbg[4 * wid + 0];
        c->viol_info[2] = a.vdbg[4 * wid + 1];
        c->viol_info[3] = a.vdbg[4 * wid + 2];
        c->viol_info[4] = h.wleaf_lo;
        c->viol_info[5] = h.wleaf_hi;
        c->viol_info[6] = h.index;
        c->viol_info[7] = h.nr;
      }
    }
    return;
  }
  // a window too large for one wave goes to a workgroup of o_big: take a queue slot BEFORE touching the state (a full queue
  // leaves the update pending for the next round)
  dev::BigJob *job = nullptr;
  if (kind_strong(kind) && h.wlen > a.big_min && a.jobs) {
    uint32_t slot = 0;
    if (lane == 0) slot = wv::atomic_add_u32(&c->njobs[par], 1u);
    slot = wv::first(slot);
    if (slot >= kBigJobs) return;
    job = &a.jobs[slot];
    if (lane == 0) c->jobs_round[par] = a.round;
  }
#if defined(PPCSR_SIM)
  if (lane == 0 && getenv("PPCSR_TRACE"))
    fprintf(stderr, "R%u commit idx=%u op=(%u,%u,%u) kind=%u index=%u gap=%u win=(%u,%u) wleaf=[%u,%u] nr=%u\n", a.round, idx, op.src,
            op.dst, op.op, kind, h.index, pl->gap, h.wstart, h.wlen, h.wleaf_lo, h.wleaf_hi, h.nr);
#endif
  dev::apply_op(a.v, op, pl, lds_wave, &a.stats[wv::block_idx() & (kStatShards - 1)], job);
  const uint32_t me1 = idx + 1u;
  if (kind_strong(kind)) {  // (a duplicate's value overwrite commutes with everything it can be reordered with: no stamp)
    const uint32_t wl = h.wleaf_lo, wh = h.wleaf_hi;
    for (uint32_t leaf = wl + (uint32_t)lane; leaf <= wh; leaf += 64) wv::atomic_max_u32(&a.wstamp[leaf], me1);
  }
  PMA_FOR_EACH_READ_LEAF_H(h, pl, lane, leaf, wv::atomic_max_u32(&a.rstamp[leaf], me1));
  if (kind_real(kind) && op.src < a.v.g.n) {
    if (lane < 2 && ((h.sdep >> lane) & 1u) && op.src + (uint32_t)lane < a.v.g.n) wv::atomic_max_u32(&a.vrs[op.src + (uint32_t)lane], me1);
    if (kind_strong(kind)) {
      const uint32_t ml = h.mv_lo, mh = h.mv_hi;
      for (uint64_t u = (uint64_t)ml + (uint64_t)lane; u <= (uint64_t)mh && ml <= mh; u += 64) wv::atomic_max_u32(&a.vws[u], me1);
    }
  }
  if (EXTRAS && (a.chain_fence & 512u)) wv::fence_mode(128u);  // (experiment: write-back after every commit of o_apply)
  if (lane == 0) a.status[wid] = OS_COMMITTED;  // (the epoch's max committed index is reduced in o_compact: a
                                                // per-update atomicMax on one word would serialise the whole round)
}


// Offsets of the buckets' lists: exclusive scan of this round's counts (ONE workgroup of 1024 threads, 32 buckets each).
PMA_KERNEL void o_bscan(OptArgs a) {
  PMA_SHARED uint32_t wtot[16];
  OptCtl *c = a.ctl;
  const uint32_t par = a.round & 1u;
  const uint32_t tid = wv::thread_idx();
  const int lane = wv::lane(), w = wv::wave_in_block();
  const uint32_t f_done = c->done, f_viol = c->violation, f_excl = c->excl, f_err = c->error;
  (void)par;
  if (!a.chain || f_done || f_viol || f_excl || f_err) return;
  constexpr uint32_t kPer = kChainBuckets / 1024u;
  uint32_t cnt[kPer], mine = 0;
#pragma unroll
  for (uint32_t q = 0; q < kPer; q++) {
    const unsigned long long x = a.bk_cnt[tid * kPer + q];
    cnt[q] = ((uint32_t)(x >> 32) == a.round) ? (uint32_t)x : 0u;
    mine += cnt[q];
  }
  uint32_t incl = mine;
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = wv::shfl(incl, lane - o < 0 ? 0 : lane - o);
    if (lane >= o) incl += y;
  }
  if (lane == 63) wtot[w] = incl;
  wv::block_sync();
  uint32_t run = incl - mine;
  for (int q = 0; q < w; q++) run += wtot[q];
#pragma unroll
  for (uint32_t q = 0; q < kPer; q++) {
    a.bk_base[tid * kPer + q] = run;
    run += cnt[q];
  }
  if (tid == 0) c->bk_round = a.round;
}

// The two halves of a chained update as CALLED device functions: o_chain with both inlined is an 11 K-instruction kernel
// whose SGPRs spill into VGPR lanes, and it came out of the compiler wrong in a way that moved with unrelated edits
// (num_neighbors short by one after some chained updates, bit-exact again with one more `if` in apply_op).  Kept small,
// the pieces are compiled like everywhere else; the call costs nothing next to the memory round trips of an update.
PMA_DEV_CALL void chain_plan(const View *v, const Op *op, Plan *plan, uint32_t r0, uint32_t r1, dev::PlanRegs *out) {
  *out = dev::plan_op_t<true>(*v, *op, plan, r0, r1);
}
PMA_DEV_CALL void chain_apply(const View *v, const Op *op, const Plan *plan, uint32_t *lds, StatShard *st) {
  dev::apply_op(*v, *op, plan, lds, st);
}

// ---- in-round chains ---------------------------------------------------------------------------------------------------
// A round commits at most one update per conflict chain: updates of one leaf (or of overlapping small windows) wait for one
// another, one round each.  Streams that keep hitting the same small neighbourhoods (a few hundred updates of ONE vertex
// whose range is a handful of leaves) are bound by exactly that: rounds = chain length.  o_chain runs after o_apply: for
// every region that still holds planned-but-uncommitted updates, the wave of the FIRST of them (stream order) goes on
// executing them one after the other — plan against the current state, check, apply — as long as
//   * the update's whole footprint (reads, writes, the sentinels it locates its range by) lies inside the region — updates
//     of different regions then cannot touch each other, and inside the region the wave itself keeps stream order;
//   * no EARLIER pending update from elsewhere reaches into the region (OptArgs::xmin, set by every deferred update for the
//     regions it touches but does not live in), none is exclusive / a barrier;
//   * no LATER update has already been committed on what it reads or writes (stamps, as everywhere).
// The first update that fails any of this ends the chain and stays pending, with everything behind it.
PMA_KERNEL void PMA_LAUNCH_BOUNDS(64, 1) o_chain(OptArgs a) {  // (launched with ONE wave per workgroup: all the registers a wave can have)
  PMA_SHARED uint32_t lds[1][3 * kLdsWindow];
  PMA_SHARED uint32_t chain_lds[1][kChainList];
  // the region being chained, staged in LDS: every chained update searches, slides and rebalances THERE (a step is a handful of
  // dependent accesses: in HBM / L2 that is ~4 us per update, and the updates of one region are sequential by definition)
  PMA_SHARED Edge reg_items[kChainRegionSlots];
  PMA_SHARED uint32_t reg_cnt[kChainRegionSlots / 2];
  OptCtl *c = a.ctl;
  const uint32_t par = a.round & 1u;
  const uint32_t gw = wv::block_idx() * (wv::block_dim() >> 6) + (uint32_t)wv::wave_in_block();  // this wave among the grid's
  const uint32_t gwaves = wv::grid_dim() * (wv::block_dim() >> 6);
  const int lane = wv::lane();
  const uint32_t f_done = c->done, f_viol = c->violation, f_excl = c->excl, f_err = c->error;
  const unsigned long long gbar = c->gbar[par], sbar = c->sbar[par];
  const uint32_t nz = c->nzones[par];
  const uint32_t sub = gw & 7u;
  uint32_t nown = c->nown[par][sub];
  if (nown > a.owners_cap) nown = a.owners_cap;
  if (!a.chain || f_done || f_viol || f_excl || f_err) return;
  if (c->bk_round != a.round) return;
  const int sh = a.v.g.sh;
  // the round's work list: one entry per bucket that has something to chain; the waves of the (small, fixed) grid share it
  for (uint32_t own = gw >> 3; own < nown; own += (gwaves >> 3)) {
  const uint32_t wid = a.owners[(uint64_t)sub * a.owners_cap + own];
  const uint32_t bucket = chain_bucket(a.bk_reg[wid]);
  const uint32_t n_l = (uint32_t)a.bk_cnt[bucket], l_base = a.bk_base[bucket];
  if (n_l > kChainList) continue;  // (longer than the LDS tile: this bucket runs without a chain)
  // the list, sorted by horizon slot (= stream order) in this wave's LDS tile
  uint32_t *ll = chain_lds[0];
  uint32_t n2 = 64;
  while (n2 < n_l) n2 <<= 1;
  for (uint32_t i = (uint32_t)lane; i < n2; i += 64u) ll[i] = i < n_l ? a.bk_list[l_base + i] : kMax;
  wv::lds_fence();
  for (uint32_t k = 2; k <= n2; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t t = (uint32_t)lane; t < (n2 >> 1); t += 64u) {
        const uint32_t lo = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), hi = lo | j;
        const bool up = (lo & k) == 0u;
        const uint32_t x = ll[lo], y = ll[hi];
        if ((x > y) == up) {
          ll[lo] = y;
          ll[hi] = x;
        }
      }
      wv::lds_fence();
    }
  const uint32_t tag = (uint32_t)(make_key(a.round, 0) >> 32);
  const uint32_t nleaves_all = (uint32_t)(a.v.g.N >> sh);
  // a chain that has ended leaves everything later of ITS region pending; a bucket holds one region unless two hash alike
  // (two are remembered; a third one ends the bucket's launch)
#define PMA_CHAIN_END(code)                                                      \
  {                                                                              \
    if (a.diag && lane == 0) wv::atomic_add_u64(&c->chain_why[code], 1ull);      \
    if (dead_a == kMax || dead_a == region) dead_a = region;                     \
    else if (dead_b == kMax || dead_b == region) dead_b = region;                \
    else all_dead = true;                                                        \
    goto next_entry;                                                             \
  }
  if (a.diag && lane == 0) wv::atomic_add_u64(&c->chain_why[9], 1ull);
  StatShard *sts = &a.stats[wv::block_idx() & (kStatShards - 1)];
  uint32_t steps = 0;
  uint32_t cur_region = kMax, dead_a = kMax, dead_b = kMax, rlo = 0, rhi = 0, rslots = 0, stop = kMax, stop_why = 0;
  bool all_dead = false, staged = false;
  View lv = a.v;
  for (uint32_t i = 0; i < n_l && !all_dead; i++) {
    const uint32_t w = ll[i];  // (horizon slot; ascending = stream order)
    const uint32_t region = a.bk_reg[w];
    const uint32_t e_st = a.status[w], nxt = a.opidx[w];
    if (e_st & OS_COMMITTED) continue;
    if (region == dead_a || region == dead_b) continue;  // (its chain has ended: everything later of that region stays pending)
    if (steps >= a.chain) {
      if (a.diag && lane == 0) wv::atomic_add_u64(&c->chain_why[2], 1ull);
      break;
    }
    if (region != cur_region) {  // (a bucket holds ONE region unless two regions hash alike)
      if (staged) {  // the region worked on so far goes back to the array
        for (uint32_t q = (uint32_t)lane; q < rslots; q += 64u) a.v.items[(uint64_t)(rlo << sh) + q] = reg_items[q];
        for (uint32_t q = (uint32_t)lane; q <= rhi - rlo; q += 64u) a.v.leafcnt[rlo + q] = reg_cnt[q];
        staged = false;
        wv::fence();
      }
      cur_region = region;
      rlo = region << a.chshift;  // first / last leaf of the region
      rhi = (rlo + (1u << a.chshift) - 1u < nleaves_all) ? rlo + (1u << a.chshift) - 1u : nleaves_all - 1u;
      rslots = (rhi - rlo + 1u) << sh;
      if (rhi + 1u >= nleaves_all || rslots > kChainRegionSlots) PMA_CHAIN_END(5)  // (the array's last region: slot N-1 has rules of its own)
      // stream index below which nothing foreign is pending: an exclusive update, a global barrier, a soft-barrier zone that
      // overlaps the region, an earlier update from elsewhere reaching into this region
      stop = kMax;
      stop_why = 0;
      const unsigned long long xm = a.xmin[region];
      if ((uint32_t)(gbar >> 32) == tag && (uint32_t)gbar < stop) { stop = (uint32_t)gbar; stop_why = 1; }
      if ((uint32_t)(sbar >> 32) == tag && (uint32_t)sbar + 1u < stop) { stop = (uint32_t)sbar + 1u; stop_why = 2; }  // (the barrier update itself may run)
      if ((uint32_t)(xm >> 32) == tag && (uint32_t)xm < stop) { stop = (uint32_t)xm; stop_why = 3; }
      for (uint32_t z = 0; z < nz && z < kMaxZones; z++) {
        const unsigned long long zk = c->zone_key[par][z];
        if ((uint32_t)(zk >> 32) == tag && (uint32_t)zk + 1u < stop && c->zone_lo[par][z] <= rhi && c->zone_hi[par][z] >= rlo) {
          stop = (uint32_t)zk + 1u;
          stop_why = 5;
        }
      }
      if (a.jobs != nullptr && c->jobs_round[par] == a.round) {
        // big windows queued by this round's o_apply are rebalanced by the workgroups of o_compact — AFTER this kernel: a region
        // such a window covers is in the middle of an update (slide and write done, rebalance pending)
        uint32_t nj = c->njobs[par];
        if (nj > kBigJobs) nj = kBigJobs;
        bool hit = false;
        for (uint32_t j = (uint32_t)lane; j < nj; j += 64) {
          const dev::BigJob jb = a.jobs[j];
          const uint32_t jl = jb.wstart >> sh, jh = (jb.wstart + jb.wlen - 1u) >> sh;
          if (jl <= rhi && jh >= rlo) hit = true;
        }
        if (wv::ballot(hit) != 0ull) PMA_CHAIN_END(8)
      }
    }
    if (nxt >= stop) {
      if (a.diag && lane == 0) wv::atomic_add_u64(&c->chain_stop[stop_why], 1ull);
      PMA_CHAIN_END(1)
    }
    if (!staged) {  // stage the region (the first update of the region that gets this far pays for it)
      for (uint32_t q = (uint32_t)lane; q < rslots; q += 64u) reg_items[q] = a.v.items[(uint64_t)(rlo << sh) + q];
      for (uint32_t q = (uint32_t)lane; q <= rhi - rlo; q += 64u) reg_cnt[q] = a.v.leafcnt[rlo + q];
      wv::lds_fence();
      lv = a.v;
      lv.items = wv::opaque_ptr((Edge *)reg_items) - (uint64_t)(rlo << sh);  // (absolute slot / leaf numbers index the staged copy)
      lv.leafcnt = wv::opaque_ptr((uint32_t *)reg_cnt) - (uint64_t)rlo;
      lv.gap_end = (uint64_t)((rhi + 1u) << sh);
      if (lv.big_window > rslots) lv.big_window = rslots;  // (no climb may look at a sibling block outside the region)
      staged = true;
    }
    {
    const Op op = a.ops[nxt];
    Plan *pl = &a.plans[w];
    // against the state as it is NOW, and from this region's slots alone (no other region is read while its owner writes it)
    dev::PlanRegs pr;
    chain_plan(&lv, &op, pl, rlo << sh, ((rhi + 1u) << sh) - 1u, &pr);
    const uint32_t kind = pr.kind;
    if (kind == K_FOREIGN) PMA_CHAIN_END(5)
    if (!kind_real(kind) || kind == K_EXCL) PMA_CHAIN_END(3)
    const bool strong = kind_strong(kind), writes = kind_writes(kind);
    if (pr.nr > 64u || pr.nlong) PMA_CHAIN_END(4)  // (read ranges beyond the register copy: leave it to the ordinary round)
    // footprint inside the region?
    bool out = false;
    const uint32_t il = pr.index >> sh;
    if (lane == 0) out = writes ? (pr.wleaf_lo < rlo || pr.wleaf_hi > rhi) : (il < rlo || il > rhi);
    if ((uint32_t)lane < pr.nr && (pr.my_lo < rlo || pr.my_hi > rhi)) out = true;
    if (lane == 1 && (pr.sdep & 1u) && (pr.sleaf_b < rlo || pr.sleaf_b > rhi)) out = true;
    if (lane == 2 && (pr.sdep & 2u) && (pr.sleaf_e < rlo || pr.sleaf_e > rhi)) out = true;
    if ((a.chain_fence & 2048u) && lane == 3 && (pr.sleaf_b < rlo || pr.sleaf_b > rhi || pr.sleaf_e < rlo || pr.sleaf_e > rhi)) out = true;  // (experiment)
    if (strong && pr.wlen > a.big_min) PMA_CHAIN_END(6)  // (a window for a workgroup: the ordinary round queues it)
    if (wv::ballot(out) != 0ull) PMA_CHAIN_END(5)
    // nothing LATER may have been committed on what this update reads or writes
    const uint32_t me1 = nxt + 1u;
    bool bad = false;
    if (strong) {
      for (uint32_t leaf = pr.wleaf_lo + (uint32_t)lane; leaf <= pr.wleaf_hi; leaf += 64)
        if (a.wstamp[leaf] > me1 || a.rstamp[leaf] > me1) bad = true;
      for (uint64_t u = (uint64_t)pr.mv_lo + (uint64_t)lane; u <= (uint64_t)pr.mv_hi && pr.mv_lo <= pr.mv_hi; u += 64)
        if (a.vrs[u] > me1 || a.vws[u] > me1) bad = true;
    } else if (kind == K_DUP) {
      if (lane == 0 && a.wstamp[pr.wleaf_lo] > me1) bad = true;
    }
    if ((uint32_t)lane < pr.nr)
      for (uint32_t leaf = pr.my_lo; leaf <= pr.my_hi; leaf++)
        if (a.wstamp[leaf] > me1) bad = true;
    if (op.src < a.v.g.n && lane < 2 && ((pr.sdep >> lane) & 1u) && op.src + (uint32_t)lane < a.v.g.n && a.vws[op.src + (uint32_t)lane] > me1) bad = true;
    if (wv::ballot(bad) != 0ull) PMA_CHAIN_END(7)  // (the ordinary round meets the same stamps and raises the violation)
    wv::fence();  // the plan record (lane 0's stores) before apply_op reads it back
#if defined(PPCSR_SIM)
    if (lane == 0 && getenv("PPCSR_TRACE"))
      fprintf(stderr, "R%u chain idx=%u op=(%u,%u,%u) kind=%u index=%u win=(%u,%u) wleaf=[%u,%u] nr=%u\n", a.round, nxt, op.src, op.dst, op.op, kind,
              pr.index, pr.wstart, pr.wlen, pr.wleaf_lo, pr.wleaf_hi, pr.nr);
#endif
    chain_apply(&lv, &op, pl, lds[0], sts);
    if (strong) {
      for (uint32_t leaf = pr.wleaf_lo + (uint32_t)lane; leaf <= pr.wleaf_hi; leaf += 64) wv::atomic_max_u32(&a.wstamp[leaf], me1);
      for (uint64_t u = (uint64_t)pr.mv_lo + (uint64_t)lane; u <= (uint64_t)pr.mv_hi && pr.mv_lo <= pr.mv_hi; u += 64) wv::atomic_max_u32(&a.vws[u], me1);
    }
    if ((uint32_t)lane < pr.nr)
      for (uint32_t leaf = pr.my_lo; leaf <= pr.my_hi; leaf++) wv::atomic_max_u32(&a.rstamp[leaf], me1);
    if (op.src < a.v.g.n && lane < 2 && ((pr.sdep >> lane) & 1u) && op.src + (uint32_t)lane < a.v.g.n) wv::atomic_max_u32(&a.vrs[op.src + (uint32_t)lane], me1);
    if (lane == 0) {
      a.status[w] = OS_CHAINED;
      wv::atomic_add_u64(&sts->chained, 1ull);
    }
      steps++;
      if (a.diag && lane == 0) wv::atomic_add_u64(&c->chain_why[10], 1ull);
      wv::fence_mode(a.chain_fence & 255u);  // this update's stores before the next one's loads
    }
  next_entry:;
  }
#undef PMA_CHAIN_END
  if (staged) {
    for (uint32_t q = (uint32_t)lane; q < rslots; q += 64u) a.v.items[(uint64_t)(rlo << sh) + q] = reg_items[q];
    for (uint32_t q = (uint32_t)lane; q <= rhi - rlo; q += 64u) a.v.leafcnt[rlo + q] = reg_cnt[q];
  }
  wv::fence();  // (the list tile and the staged region are reused by this wave's next bucket)
  }
}

// stable compaction of the deferred updates into the next carry list + next round's bookkeeping
// ONE workgroup; everything it needs is requested in one batch of independent loads (control block, then each thread's
// run of statuses and indices), because at ~6 K entries this step is nothing but load latency.
// kC: 64-slot chunks per wave held in registers (covers a horizon of kC * blockDim).  wsum: 16 words of LDS, s_first_p: 1.
template <uint32_t kC>
PMA_DEV void compact_block(const OptArgs &a, uint32_t *wsum, uint32_t *s_first_p) {
  OptCtl *c = a.ctl;
  const uint32_t par = a.round & 1u;
  const uint32_t f_done = c->done, f_viol = c->violation, f_excl = c->excl, f_err = c->error;
  const uint32_t hor = c->hor[par], cn = c->carry_n[par], nf = c->next_fresh[par];
  const uint32_t cur_h = c->cur_horizon, max_h = c->max_horizon, wcap = c->width_cap, adaptive = c->adaptive, e1 = c->e1;
  const unsigned long long gb = c->gbar[par];
  const unsigned long long n_rounds = c->rounds, n_committed = c->committed, n_planned = c->planned;
  if (f_done || f_viol || f_excl || f_err) return;
  const uint32_t used = cn < hor ? cn : hor;
  const uint32_t *cin = par ? a.carry1 : a.carry0;
  uint32_t *cout = par ? a.carry0 : a.carry1;
  const uint32_t tid = wv::thread_idx(), bd = wv::block_dim();
  const int lane = wv::lane(), w = wv::wave_in_block();
  const uint32_t nw = bd >> 6;
  // Every wave owns a run of consecutive 64-slot chunks (lane l of chunk c: slot wbase + 64 c + l, so every load and every
  // store is coalesced — with a run of consecutive slots per THREAD the 2 x kC loads of a wave touched 64 lines each, and
  // at 18 K entries one CU's address path made this 15 us); a ballot per chunk counts and ranks, the waves' totals go
  // through LDS.
  const uint32_t cpw = (hor + nw * 64u - 1u) / (nw * 64u);  // chunks per wave
  const uint32_t wbase = (uint32_t)w * cpw * 64u;
  const uint64_t lt = (1ull << lane) - 1ull;
  uint32_t wkeep = 0, mymaxc = 0;
  uint32_t st[kC], oi[kC];
  const bool regs = cpw <= kC;
  if (regs) {
#pragma unroll
    for (uint32_t q = 0; q < kC; q++) {
      const uint32_t sl = wbase + q * 64u + (uint32_t)lane;
      const bool in = q < cpw && sl < hor;
      st[q] = in ? a.status[sl] : OS_COMMITTED;
      oi[q] = in ? a.opidx[sl] : 0u;
    }
#pragma unroll
    for (uint32_t q = 0; q < kC; q++) {
      if (q >= cpw) break;  // (wave-uniform)
      const bool in = wbase + q * 64u + (uint32_t)lane < hor;
      wkeep += (uint32_t)wv::popc64(wv::ballot(in && !(st[q] & OS_COMMITTED)));
      if (in && (st[q] & OS_COMMITTED) && oi[q] + 1u > mymaxc) mymaxc = oi[q] + 1u;
    }
  } else {
    for (uint32_t q = 0; q < cpw; q++) {
      const uint32_t sl = wbase + q * 64u + (uint32_t)lane;
      const bool in = sl < hor;
      const uint32_t s1 = in ? a.status[sl] : OS_COMMITTED, x = in ? a.opidx[sl] : 0u;
      wkeep += (uint32_t)wv::popc64(wv::ballot(in && !(s1 & OS_COMMITTED)));
      if (in && (s1 & OS_COMMITTED) && x + 1u > mymaxc) mymaxc = x + 1u;
    }
  }
  if (lane == 0) wsum[w] = wkeep;
  if (tid == 0) *s_first_p = kMax;
  wv::block_sync();
  uint32_t woff = 0, tot = 0;
  for (uint32_t q = 0; q < nw; q++) {
    if (q < (uint32_t)w) woff += wsum[q];
    tot += wsum[q];
  }
  uint32_t o = woff;
  if (regs) {
#pragma unroll
    for (uint32_t q = 0; q < kC; q++) {
      if (q >= cpw) break;
      const bool keep = wbase + q * 64u + (uint32_t)lane < hor && !(st[q] & OS_COMMITTED);
      const uint64_t m = wv::ballot(keep);
      if (keep) {
        const uint32_t pos = o + (uint32_t)wv::popc64(m & lt);
        if (pos == 0) *s_first_p = oi[q];
        cout[pos] = oi[q];
      }
      o += (uint32_t)wv::popc64(m);
    }
  } else {
    for (uint32_t q = 0; q < cpw; q++) {
      const uint32_t sl = wbase + q * 64u + (uint32_t)lane;
      const bool in = sl < hor;
      const uint32_t s1 = in ? a.status[sl] : OS_COMMITTED, x = in ? a.opidx[sl] : 0u;
      const bool keep = in && !(s1 & OS_COMMITTED);
      const uint64_t m = wv::ballot(keep);
      if (keep) {
        const uint32_t pos = o + (uint32_t)wv::popc64(m & lt);
        if (pos == 0) *s_first_p = x;
        cout[pos] = x;
      }
      o += (uint32_t)wv::popc64(m);
    }
  }
  const uint32_t ncommitted = hor - tot;
  const uint32_t kept = tot;
  {  // one atomic per wave: a thousand same-address atomics would serialise in L2 for longer than the rest of this kernel
    uint32_t wmax = mymaxc;
    for (int o2 = 32; o2 > 0; o2 >>= 1) {
      const uint32_t y = wv::shfl(wmax, lane ^ o2);
      wmax = y > wmax ? y : wmax;
    }
    if (lane == 0 && wmax) wv::atomic_max_u32(&c->maxc, wmax);
  }
  for (uint32_t i = used + tid; i < cn; i += bd) {  // carry entries beyond the horizon
    const uint32_t x = cin[i];
    if (kept + (i - used) == 0) *s_first_p = x;
    cout[kept + (i - used)] = x;
  }
  wv::block_sync();
  if (tid == 0) {
    const uint32_t new_cn = kept + (cn - used);
    const uint32_t new_nf = nf + (hor - used);
    // adaptive width: dependency chains bound the number of commits per round (a hot vertex whose range sits at its
    // density bounds yields a few hundred disjoint windows per round however many updates are planned), so planning far
    // more than can commit only makes every round slower — but a round's time grows far slower than its width (~30 us
    // + ~2.5 us per 1024 updates), so width is only given up when almost nothing of it commits.  Narrow by 1/4 when less
    // than 10 % of a full-width round committed, widen by 1/4 when more than 30 % did.
    // Above one chip-full of waves (`resident`) only whole multiples make sense (a partly filled second pass costs a
    // full pass of latency), and a multiple is only worth its re-planning when nearly all of the round commits: up at
    // > 85 %, back down at < 70 % (config #4's partitions at critical density commit 60-70 % of a chip-full: at twice the
    // width they lost 5 %; configs #2 / #3 commit 93-97 % and gain 11-12 %).
    uint32_t ch = cur_h ? cur_h : wcap;
    const uint32_t res = c->resident;
    if (adaptive && hor >= ch) {  // only full-width rounds carry information about the width
      if (res && ch >= res) {
        if (ncommitted * 100u > hor * 85u) ch += res;
        else if (ch > res && ncommitted * 100u < hor * 70u) ch -= res;
        else if (ch == res && ncommitted * 100u < hor * 10u) ch -= ch / 4u;
      } else {
        if (ncommitted * 100u > hor * 30u) ch += ch / 4u;
        else if (ncommitted * 100u < hor * 10u) ch -= ch / 4u;
        if (res && ch > res) ch = res;
      }
    }
    if (res && ch > res) ch -= ch % res;
    if (ch < 1024u) ch = 1024u;
    if (ch > wcap) ch = wcap;  // (the launch grid — max_h — bounds the next round below, not the adapted width itself: the
                               // host narrows the grid at the tail of an epoch)
    c->cur_horizon = ch;
    uint32_t nh = new_cn + (e1 - new_nf);
    if (nh > ch) nh = ch;
    if (nh > max_h) nh = max_h;
    c->carry_n[par ^ 1u] = new_cn;
    c->next_fresh[par ^ 1u] = new_nf;
    c->hor[par ^ 1u] = nh;
    c->gbar[par ^ 1u] = ~0ull;
    c->gbar[par] = ~0ull;
    c->sbar[par ^ 1u] = ~0ull;
    c->sbar[par] = ~0ull;
    c->nzones[par ^ 1u] = 0u;
    c->nzones[par] = 0u;
    for (int q = 0; q < 8; q++) c->nown[par ^ 1u][q] = c->nown[par][q] = 0u;
    c->njobs[par ^ 1u] = 0;
    c->skip = kMax;
    const bool done = (new_cn == 0 && new_nf == e1);
    if (done) c->done = 1;
    const uint32_t lowest = new_cn ? *s_first_p : new_nf;
    const uint32_t tag = (uint32_t)(make_key(a.round, 0) >> 32);
    c->resume_par = par ^ 1u;
    if (!done && (uint32_t)(gb >> 32) == tag && (uint32_t)gb == lowest) {
      c->excl = 1;
      c->excl_idx = lowest;
    }
    if (n_rounds < 96) {
      c->hist[2 * n_rounds] = hor;
      c->hist[2 * n_rounds + 1] = ncommitted;
    }
    c->rounds = n_rounds + 1ull;
    c->committed = n_committed + (unsigned long long)ncommitted;
    c->planned = n_planned + (unsigned long long)hor;
  }
}

// (Forcing 8 waves per SIMD — __launch_bounds__(256, 8) on o_plan / o_apply, a 256-slot LDS tile — for 8192-wide rounds was
// measured again in round 2: 56 / 44 B of scratch per lane and 113-123 M updates/s against 141 at 6 waves per SIMD.)
PMA_KERNEL void o_apply(OptArgs a) {
  PMA_SHARED uint32_t lds[4][3 * kLdsWindow];
  o_apply_wave<false>(a, lds[wv::wave_in_block()]);
}
PMA_KERNEL void o_apply_x(OptArgs a) {
  PMA_SHARED uint32_t lds[4][3 * kLdsWindow];
  o_apply_wave<true>(a, lds[wv::wave_in_block()]);
}

// (Folding the compaction into o_apply's last-finishing workgroup was measured and dropped: the device-scope fences the
// ticket needs make every workgroup write back its XCD's L2, and the round got 3x slower than with a separate launch.)
// markers: empty kernels with distinct names; `set_option("marker", i)` launches k_mark_<i> on the engine's stream so that
// tools/roofline_summary.py can cut sections (timed region, one isolated rebalance, one scan) out of a rocprofv3 kernel
// trace / counter collection of bench.py
#define PMA_MARK(i) PMA_KERNEL void k_mark_##i(uint32_t *p) { if (p && wv::thread_idx() == 0xFFFFFFFFu) *p = i; }
PMA_MARK(0) PMA_MARK(1) PMA_MARK(2) PMA_MARK(3) PMA_MARK(4) PMA_MARK(5) PMA_MARK(6) PMA_MARK(7)
#undef PMA_MARK

// test hook: one workgroup rebalances one window with the big-window routine (leaf counts must be exact)
PMA_KERNEL void k_block_rebalance(View v, uint64_t wstart, uint64_t wlen, Edge *scratch) {
  PMA_SHARED dev::BigShared sh;
  dev::redistribute_block(v, wstart, wlen, scratch, sh);
}

// Workgroup 0: the compaction.  Workgroups 1 .. : the round's queued big-window rebalances, one workgroup per window
// (dev::redistribute_block) — independent of the compaction (they only finish the rebalance of updates that have already
// been committed), so they share its launch instead of paying a kernel boundary of their own.
PMA_KERNEL void o_compact(OptArgs a) {
  PMA_SHARED uint32_t wsum[16];
  PMA_SHARED uint32_t s_first;
  PMA_SHARED dev::BigShared sh;
  if (wv::block_idx() == 0) {
    compact_block<24>(a, wsum, &s_first);  // (24 x 1024 threads: rounds up to 24576 wide stay in registers)
    return;
  }
  OptCtl *c = a.ctl;
  const uint32_t par = a.round & 1u;
  // (NOT c->done / c->excl: workgroup 0 sets them during this very launch.  A violation rolls the epoch back anyway.)
  const uint32_t f_viol = c->violation, f_err = c->error;
  uint32_t nj = c->njobs[par];
  if (f_viol || f_err || nj == 0 || c->jobs_round[par] != a.round) return;
  if (nj > kBigJobs) nj = kBigJobs;
  const uint32_t nwg = wv::grid_dim() - 1u, me = wv::block_idx() - 1u;
  for (uint32_t jb = me; jb < nj; jb += nwg) {
    const dev::BigJob job = a.jobs[jb];
    dev::redistribute_block(a.v, job.wstart, job.wlen, a.bigscratch + (uint64_t)me * a.bigscratch_stride, sh);
    wv::block_sync();  // the shared prefix / table are reused by the next job
  }
}

// ---- owner bucketing for the multi-GPU exchange (PPPCSR routing rule, PPPCSR.cpp:46-66) ----------------------------
// Stable counting sort of a block of the update stream by owning partition, with `src` made partition-local.
// Tile = 2048 updates per workgroup; (1) per-tile histogram, (2) one small scan (partition-major, tile-minor),
// (3) scatter: rank inside the wave from ballots over the distinct owners present, inside the tile from an LDS
// prefix over (row, wave), across tiles from the scan — so every bucket keeps stream order.
constexpr uint32_t kBucketRows = 8;                          // rows of 256 updates per tile
constexpr uint32_t kBucketTile = 256 * kBucketRows;
constexpr uint32_t kMaxParts = 64;
struct PartTable {  // first global vertex of every partition (PPPCSR::distribution, PPPCSR.h:57), passed by value
  uint32_t start[kMaxParts];
};
// PPPCSR::get_partiton (PPPCSR.cpp:58-66): the last partition whose first vertex is <= src (starts are non-decreasing,
// start[0] = 0; equal starts — empty partitions — resolve to the last of them, as the reference's linear walk does)
PMA_DEV uint32_t owner_of_src(uint32_t src, const uint32_t *pstart /* LDS */, uint32_t nparts) {
  uint32_t lo = 0, hi = nparts;
  while (hi - lo > 1u) {
    const uint32_t mid = (lo + hi) >> 1;
    if (pstart[mid] <= src) lo = mid; else hi = mid;
  }
  return lo;
}
// counts[w][p] for the 4 waves of one row; returns this lane's rank among same-owner lanes of its wave
PMA_DEV uint32_t bucket_rank_in_wave(uint32_t owner, bool valid, uint32_t *wave_counts /* [kMaxParts] of this wave */) {
  const int lane = wv::lane();
  uint64_t remaining = wv::ballot(valid);
  uint32_t myrank = 0;
  while (remaining) {
    const int l0 = wv::ctz64(remaining);
    const uint32_t p0 = wv::shfl(owner, l0);
    const uint64_t m = wv::ballot(valid && owner == p0);
    if (valid && owner == p0) myrank = dev::lanemask_lt_count(m, lane);
    if (lane == 0) wave_counts[p0] = (uint32_t)wv::popc64(m);
    remaining &= ~m;
  }
  return myrank;
}
PMA_KERNEL void k_bucket_hist(const Op *ops, uint64_t n, PartTable tab, uint32_t nparts, uint32_t *hist /* [ntiles][nparts] */) {
  PMA_SHARED uint32_t cnt[kBucketRows][4][kMaxParts];
  PMA_SHARED uint32_t pstart[kMaxParts];
  const uint32_t tid = wv::thread_idx();
  const int w = wv::wave_in_block();
  const uint64_t tile = wv::block_idx();
  for (uint32_t i = tid; i < kBucketRows * 4 * kMaxParts; i += 256) (&cnt[0][0][0])[i] = 0;
  if (tid < kMaxParts) pstart[tid] = tab.start[tid];
  wv::block_sync();
  for (uint32_t r = 0; r < kBucketRows; r++) {
    const uint64_t i = tile * kBucketTile + (uint64_t)r * 256 + tid;
    const bool valid = i < n;
    const uint32_t owner = valid ? owner_of_src(ops[i].src, pstart, nparts) : 0u;
    (void)bucket_rank_in_wave(owner, valid, cnt[r][w]);
  }
  wv::block_sync();
  for (uint32_t p = tid; p < nparts; p += 256) {
    uint32_t t = 0;
    for (uint32_t r = 0; r < kBucketRows; r++)
      for (uint32_t q = 0; q < 4; q++) t += cnt[r][q][p];
    hist[tile * nparts + p] = t;
  }
}
// exclusive offsets, partition-major: off[tile][p] = sum_{p'<p} total[p'] + sum_{tile'<tile} hist[tile'][p]; counts[p] = total[p]
// ONE workgroup of kBucketScanThreads = 16 groups x 64 partitions: group g owns a contiguous run of tiles, thread (g, p)
// sums partition p over that run, the 16 x 64 partial sums are scanned in LDS, and the run is walked once more to write
// the offsets.  (One thread per partition walking all tiles — 4883 of them for a 10 M-update block — took 1.2 ms.)
constexpr uint32_t kBucketScanThreads = 1024;
PMA_KERNEL void k_bucket_scan(uint32_t *hist, uint64_t ntiles, uint32_t nparts, unsigned long long *counts) {
  constexpr uint32_t G = kBucketScanThreads / kMaxParts;
  PMA_SHARED unsigned long long part[G][kMaxParts];  // sum of partition p over group g's tiles
  PMA_SHARED unsigned long long tot[kMaxParts];
  const uint32_t tid = wv::thread_idx();
  const uint32_t g = tid / kMaxParts, p = tid % kMaxParts;
  const uint64_t per = (ntiles + G - 1) / G;
  const uint64_t t0 = (uint64_t)g * per, t1 = (t0 + per < ntiles) ? t0 + per : ntiles;
  unsigned long long mine = 0;
  if (p < nparts)
    for (uint64_t t = t0; t < t1; t++) mine += hist[t * nparts + p];
  part[g][p] = mine;
  wv::block_sync();
  if (g == 0 && p < nparts) {
    unsigned long long run = 0;
    for (uint32_t q = 0; q < G; q++) {
      const unsigned long long x = part[q][p];
      part[q][p] = run;  // partition p: what the groups in front of q hold
      run += x;
    }
    tot[p] = run;
    counts[p] = run;
  }
  wv::block_sync();
  if (p < nparts) {
    unsigned long long run = part[g][p];
    for (uint32_t q = 0; q < p; q++) run += tot[q];  // + everything of the partitions in front of p
    for (uint64_t t = t0; t < t1; t++) {
      const uint32_t c = hist[t * nparts + p];
      hist[t * nparts + p] = (uint32_t)run;
      run += c;
    }
  }
}
PMA_KERNEL void k_bucket_scatter(const Op *ops, uint64_t n, PartTable tab, uint32_t nparts, const uint32_t *off, Op *out) {
  PMA_SHARED uint32_t cnt[kBucketRows][4][kMaxParts];
  PMA_SHARED uint32_t pstart[kMaxParts];
  const uint32_t tid = wv::thread_idx();
  const int w = wv::wave_in_block();
  const uint64_t tile = wv::block_idx();
  for (uint32_t i = tid; i < kBucketRows * 4 * kMaxParts; i += 256) (&cnt[0][0][0])[i] = 0;
  if (tid < kMaxParts) pstart[tid] = tab.start[tid];
  wv::block_sync();
  Op mine[kBucketRows];
  uint32_t owner[kBucketRows], rank[kBucketRows];
  for (uint32_t r = 0; r < kBucketRows; r++) {
    const uint64_t i = tile * kBucketTile + (uint64_t)r * 256 + tid;
    const bool valid = i < n;
    mine[r] = valid ? ops[i] : Op{0u, 0u, 0u};
    owner[r] = valid ? owner_of_src(mine[r].src, pstart, nparts) : 0u;
    rank[r] = bucket_rank_in_wave(owner[r], valid, cnt[r][w]);
  }
  wv::block_sync();
  // exclusive prefix over (row, wave) per partition, in place (one thread per partition)
  for (uint32_t p = tid; p < nparts; p += 256) {
    uint32_t run = 0;
    for (uint32_t r = 0; r < kBucketRows; r++)
      for (uint32_t q = 0; q < 4; q++) {
        const uint32_t c = cnt[r][q][p];
        cnt[r][q][p] = run;
        run += c;
      }
  }
  wv::block_sync();
  for (uint32_t r = 0; r < kBucketRows; r++) {
    const uint64_t i = tile * kBucketTile + (uint64_t)r * 256 + tid;
    if (i < n) {
      const uint32_t p = owner[r];
      Op o = mine[r];
      o.src = o.src - pstart[p];  // partition-local source, global destination (PPPCSR.cpp:46-52)
      out[(uint64_t)off[tile * nparts + p] + cnt[r][w][p] + rank[r]] = o;
    }
  }
}

}  // namespace ppcsr
